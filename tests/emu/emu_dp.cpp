// emu_dp.cpp -- HOST PHASE EMULATOR of the structured-DP workgroup.  TEST INFRASTRUCTURE ONLY.
//
// There is no GPU in the build container, so the CPU test-suite executes the *same* per-thread
// kernel bodies (vlgae_amd/csrc/vlg_dp_core.h: dmv_run / dep_run) with `nt` host threads standing
// in for the lanes of one workgroup.  __syncthreads() is replaced by a token barrier that also
// serialises the threads inside every barrier-delimited phase in a chosen order (forward,
// reverse or shuffled).  A phase whose result depends on that order has an intra-phase race;
// the tests require bit-identical results across orders and parity with the oracle.
//
// This is NOT a CPU fallback: nothing in vlgae_amd/ loads it, and the product raises if the
// HIP library is missing.
#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <mutex>
#include <random>
#include <thread>
#include <vector>

#include "../../vlgae_amd/csrc/vlg_dp_core.h"

namespace {

struct Token {
    std::mutex mu;
    std::condition_variable cv;
    long ticket = 0;
    int nt = 1;
    void wait_for(long t) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return ticket == t; });
    }
    void advance() {
        {
            std::lock_guard<std::mutex> lk(mu);
            ++ticket;
        }
        cv.notify_all();
    }
};

std::vector<int> make_order(int nt, int order) {
    std::vector<int> rank(nt);
    for (int t = 0; t < nt; ++t) rank[t] = t;
    if (order == 1) std::reverse(rank.begin(), rank.end());
    if (order >= 2) { std::mt19937 g(order); std::shuffle(rank.begin(), rank.end(), g); }
    return rank;   // rank[tid] = position of tid inside every phase
}

// Host stand-in for the device's cross-lane policy (DevX in vlg_dp.hip).  sync() is the token barrier;
// the group all-reduces go through a shared exchange buffer and replay the device's butterfly tree
// (lane t combines with lane t^1, then t^2, ...) so that host and device results agree bit for bit.
struct HostX {
    static constexpr bool kSkipDeadWaves = false;   // here an all-reduce is a barrier: every lane takes part
    static constexpr bool kChartsInLds = false;     // heap arrays: out-of-range split points keep their clamped loads
    Token* tok;
    int nt, tid, rho;
    long phase = 0;
    float* xf;   // [nt][8]
    int* xi;     // [nt][8]

    void lockstep() { sync(); }   // device: lanes of a group share a wavefront (see DevX::lockstep)
    bool uniform(bool v) { return v; }
    int uniform(int v) { return v; }
    void sync() {
        tok->advance();
        ++phase;
        tok->wait_for(phase * nt + rho);
    }
    template <int n, typename Op>
    void butterfly(float* v, int* a, int G, Op op) {
        if (G == 1) return;   // uniform over the workgroup
        for (int k = 0; k < n; ++k) { xf[tid * 8 + k] = v[k]; if (a) xi[tid * 8 + k] = a[k]; }
        sync();
        const int base = tid & ~(G - 1);
        for (int k = 0; k < n; ++k) {
            float cur[64], nxt[64];
            int ci[64], ni[64];
            for (int t = 0; t < G; ++t) { cur[t] = xf[(base + t) * 8 + k]; ci[t] = a ? xi[(base + t) * 8 + k] : 0; }
            for (int s = 1; s < G; s <<= 1) {
                for (int t = 0; t < G; ++t) op(cur[t], ci[t], cur[t ^ s], ci[t ^ s], nxt[t], ni[t]);
                for (int t = 0; t < G; ++t) { cur[t] = nxt[t]; ci[t] = ni[t]; }
            }
            v[k] = cur[tid - base];
            if (a) a[k] = ci[tid - base];
        }
        sync();
    }
    template <int n>
    void allreduce_max(float* v, int G) {
        butterfly<n>(v, nullptr, G, [](float x, int, float y, int, float& o, int& oi) { o = fmaxf(x, y); oi = 0; });
    }
    template <int n>
    void allreduce_sum(float* v, int G) {
        butterfly<n>(v, nullptr, G, [](float x, int, float y, int, float& o, int& oi) { o = x + y; oi = 0; });
    }
    // true broadcasts, like the device's ds_bpermute: only the source lane's value is ever read
    template <int n>
    void bcast(float* v, int G, const int* src) {
        if (G == 1) return;
        for (int k = 0; k < n; ++k) xf[tid * 8 + k] = v[k];
        sync();
        const int base = tid & ~(G - 1);
        for (int k = 0; k < n; ++k) v[k] = xf[(base + src[k]) * 8 + k];
        sync();
    }
    void group_bcast4(float* v, int G, int s01, int s23) { const int src[4] = {s01, s01, s23, s23}; bcast<4>(v, G, src); }
    void group_bcast2(float* v, int G, int s0, int s1) { const int src[2] = {s0, s1}; bcast<2>(v, G, src); }
    template <int n>
    void allreduce_argmax(float* v, int* a, int G) {
        butterfly<n>(v, a, G, [](float x, int xa, float y, int ya, float& o, int& oi) {
            const bool take = y > x || (y == x && ya < xa);
            o = take ? y : x;
            oi = take ? ya : xa;
        });
    }
};

template <typename Body>
void run_workgroup(int nt, int order, Body body) {
    Token tok;
    tok.nt = nt;
    const std::vector<int> rank = make_order(nt, order);
    std::vector<float> xf((size_t)nt * 8);
    std::vector<int> xi((size_t)nt * 8);
    std::vector<std::thread> th;
    for (int tid = 0; tid < nt; ++tid)
        th.emplace_back([&, tid] {
            HostX x;
            x.tok = &tok; x.nt = nt; x.tid = tid; x.rho = rank[tid]; x.xf = xf.data(); x.xi = xi.data();
            tok.wait_for(x.rho);
            body(tid, x);
            tok.advance();
        });
    for (auto& t : th) t.join();
}

// The working set is carved out of ONE buffer with exactly the kernel's layout (DmvLayout / DepLayout),
// followed by a canary: a phase body that writes an array the layout did not allocate (e.g. the
// outside-pass tape when only the inside pass was requested) trips it.
constexpr size_t kCanary = 1 << 16;
static int g_canary_trips = 0;
static int g_mode = 0;   // DMV placement mode to emulate (0: one carve; 1..3: part of the working set in a second arena = the workspace)

struct Arena {
    std::vector<unsigned char> buf;
    size_t total;
    explicit Arena(size_t total_) : buf(total_ + kCanary, 0xA5), total(total_) {}
    char* at(size_t off) { return reinterpret_cast<char*>(buf.data()) + off; }
    void check() {
        for (size_t i = total; i < buf.size(); ++i)
            if (buf[i] != 0xA5) { ++g_canary_trips; return; }
    }
};

template <int SR, bool BWD, typename In>
void emu_dmv_one(const typename In::T* dec, const typename In::T* attach, int len, int N, float glogZ, float* logZ,
                 float* gdec, float* gatt, int nt, int order, long long* heads = nullptr) {
    const bool walk = BWD && SR == VLG_SR_MAX;   // the launcher's rule (vlg_dp.hip: run_dmv)
    const vlg::DmvLayout L(N, BWD, SR == VLG_SR_MAX, g_mode, walk);   // mode 0: one contiguous carve, like LDS
    Arena A(L.lds_bytes), W(L.ws_bytes);
    auto at = [&](const vlg::Region& r) { return r.lds ? A.at(r.off) : W.at(r.off); };
    vlg::DmvCtx c;
    c.walk = walk;
    c.Ne = len + 1; c.len = len; c.P = vlg::chart_pitch(N);
    c.C = (float2*)at(L.C_in); c.I = (float2*)at(L.I_in); c.C2 = (float2*)at(L.C); c.I2 = (float2*)at(L.I);
    c.S = (float*)at(L.S);
    c.bpS = (unsigned char*)at(L.bpS); c.bpC = (unsigned char*)at(L.bpC);
    c.gCc = (float*)at(L.gCc); c.gCi = (float2*)at(L.gCi); c.gI = (float2*)at(L.gI);
    c.decs = (float*)at(L.decs); c.gdecs = (float*)at(L.gdecs);
    vlg::MergedIO<In> io;
    io.dec = dec; io.attach = attach; io.N = N; io.gdec = gdec; io.gatt = gatt; io.heads = heads;
    run_workgroup(nt, order, [&](int tid, HostX& x) {
        // the long-sentence placements run the chunked form of the long spans (vlg_dp.hip: dmv_run<SR, BWD, MODE != 0>)
        if (g_mode != 0) vlg::dmv_run<SR, BWD, true>(c, io, glogZ, logZ, tid, nt, x);
        else vlg::dmv_run<SR, BWD, false>(c, io, glogZ, logZ, tid, nt, x);
    });
    A.check();
    W.check();
}

// rule-table input (RuleIO): one sentence
template <int SR, bool BWD>
void emu_rules_one(const vlg::RuleIO<vlg::F32In>& io, int len, float glogZ, float* logZ, int nt, int order) {
    const int N = io.L + 1;
    const bool walk = BWD && SR == VLG_SR_MAX;
    const vlg::DmvLayout L(N, BWD, SR == VLG_SR_MAX, 0, walk);
    Arena A(L.lds_bytes);
    vlg::DmvCtx c;
    c.walk = walk;
    c.Ne = len + 1; c.len = len; c.P = vlg::chart_pitch(N);
    c.C = (float2*)A.at(L.C.off); c.I = (float2*)A.at(L.I.off); c.C2 = c.C; c.I2 = c.I; c.S = (float*)A.at(L.S.off);
    c.bpS = (unsigned char*)A.at(L.bpS.off); c.bpC = (unsigned char*)A.at(L.bpC.off);
    c.gCc = (float*)A.at(L.gCc.off); c.gCi = (float2*)A.at(L.gCi.off); c.gI = (float2*)A.at(L.gI.off);
    c.decs = (float*)A.at(L.decs.off); c.gdecs = (float*)A.at(L.gdecs.off);
    run_workgroup(nt, order, [&](int tid, HostX& x) { vlg::dmv_run<SR, BWD>(c, io, glogZ, logZ, tid, nt, x); });
    A.check();
}

template <int SR, bool BWD, typename In>
void emu_dep_one(const typename In::T* arc, int len, int N, float glogZ, float* logZ, float* garc, int nt, int order,
                 long long* heads = nullptr) {
    const vlg::DepLayout L(N, BWD, SR == VLG_SR_MAX, 0);
    Arena A(L.lds_bytes);
    vlg::DepCtx c;
    c.Ne = len + 1; c.len = len; c.P = vlg::chart_pitch(N);
    c.C = (float*)A.at(L.C.off); c.I = (float*)A.at(L.I.off); c.S = (float*)A.at(L.S.off);
    c.bpS = (unsigned char*)A.at(L.bpS.off); c.bpC = (unsigned char*)A.at(L.bpC.off);
    c.gCc = (float*)A.at(L.gCc.off); c.gCi = (float*)A.at(L.gCi.off); c.gI = (float*)A.at(L.gI.off);
    run_workgroup(nt, order, [&](int tid, HostX& x) {
        vlg::dep_run<SR, BWD, In>(c, arc, N, glogZ, logZ, garc, heads, tid, nt, x);
    });
    A.check();
}

template <typename In>
int dmv_batch(const void* dec_, const void* attach_, const int64_t* lengths, int B, int N, int semiring,
              const float* glogZ, float* logZ, float* gdec, float* gatt, int nt, int order) {
    auto dec = (const typename In::T*)dec_;
    auto attach = (const typename In::T*)attach_;
    for (int b = 0; b < B; ++b) {
        const int len = (int)lengths[b];
        const size_t doff = (size_t)b * N * 8, aoff = (size_t)b * N * N * 2;
        if (len < 1 || len > N - 1) return -1;
        const float g = glogZ ? glogZ[b] : 1.f;
        if (gdec) {
            if (semiring == 0) emu_dmv_one<0, true, In>(dec + doff, attach + aoff, len, N, g, logZ + b, gdec + doff, gatt + aoff, nt, order);
            else emu_dmv_one<1, true, In>(dec + doff, attach + aoff, len, N, g, logZ + b, gdec + doff, gatt + aoff, nt, order);
        } else {
            if (semiring == 0) emu_dmv_one<0, false, In>(dec + doff, attach + aoff, len, N, g, logZ + b, nullptr, nullptr, nt, order);
            else emu_dmv_one<1, false, In>(dec + doff, attach + aoff, len, N, g, logZ + b, nullptr, nullptr, nt, order);
        }
    }
    return 0;
}

template <typename In>
int dep_batch(const void* arc_, const int64_t* lengths, int B, int N, int semiring, const float* glogZ, float* logZ,
              float* garc, int nt, int order) {
    auto arc = (const typename In::T*)arc_;
    for (int b = 0; b < B; ++b) {
        const int len = lengths ? (int)lengths[b] : N - 1;
        const size_t off = (size_t)b * N * N;
        if (len < 1 || len > N - 1) return -1;
        const float g = glogZ ? glogZ[b] : 1.f;
        if (garc) {
            if (semiring == 0) emu_dep_one<0, true, In>(arc + off, len, N, g, logZ + b, garc + off, nt, order);
            else emu_dep_one<1, true, In>(arc + off, len, N, g, logZ + b, garc + off, nt, order);
        } else {
            if (semiring == 0) emu_dep_one<0, false, In>(arc + off, len, N, g, logZ + b, nullptr, nt, order);
            else emu_dep_one<1, false, In>(arc + off, len, N, g, logZ + b, nullptr, nt, order);
        }
    }
    return 0;
}

}  // namespace

extern "C" {
int emu_canary_trips(void) { return g_canary_trips; }
void emu_set_dmv_mode(int mode) { g_mode = mode; }
// rule-table entry (f32): grads must be zero-filled by the caller; any of them may be null together (inside only)
int emu_dmv1o_rules(const float* rule, const float* dec, const float* root, int root_per_sentence, const int64_t* token,
                    const unsigned char* head_mask, const int64_t* lengths, int B, int L, int T, int semiring, float fill,
                    float* logZ, float* g_rule, float* g_dec, float* g_root, long long* heads, int nt, int order) {
    for (int b = 0; b < B; ++b) {
        vlg::RuleIO<vlg::F32In> io;
        io.rule = rule + (size_t)b * L * T * 4; io.dec = dec + (size_t)b * L * 8;
        io.root = root + (root_per_sentence ? (size_t)b * T : 0);
        io.token = (const long long*)token + (size_t)b * L;
        io.head_mask = head_mask ? head_mask + (size_t)b * L : nullptr;
        io.L = L; io.T = T; io.fill = fill;
        io.g_rule = g_rule ? g_rule + (size_t)b * L * T * 4 : nullptr;
        io.g_dec = g_dec ? g_dec + (size_t)b * L * 8 : nullptr;
        io.g_root = g_root ? g_root + (size_t)b * T : nullptr;
        io.heads = heads ? heads + (size_t)b * (L + 1) : nullptr;
        const bool bwd = g_rule || heads;
        const int len = (int)lengths[b];
        if (semiring == 0) { if (bwd) emu_rules_one<0, true>(io, len, 1.f, logZ + b, nt, order); else emu_rules_one<0, false>(io, len, 1.f, logZ + b, nt, order); }
        else { if (bwd) emu_rules_one<1, true>(io, len, 1.f, logZ + b, nt, order); else emu_rules_one<1, false>(io, len, 1.f, logZ + b, nt, order); }
    }
    return 0;
}
// decode mode: Max semiring, heads out, no gradient buffers (f32 inputs)
int emu_dmv1o_decode(const float* dec, const float* attach, const int64_t* lengths, int B, int N, float* best,
                     long long* heads, int nt, int order) {
    for (int b = 0; b < B; ++b)
        emu_dmv_one<1, true, vlg::F32In>(dec + (size_t)b * N * 8, attach + (size_t)b * N * N * 2, (int)lengths[b], N, 1.f,
                                         best + b, nullptr, nullptr, nt, order, heads + (size_t)b * N);
    return 0;
}
int emu_deptree_decode(const float* arc, const int64_t* lengths, int B, int N, float* best, long long* heads, int nt,
                       int order) {
    for (int b = 0; b < B; ++b)
        emu_dep_one<1, true, vlg::F32In>(arc + (size_t)b * N * N, (int)lengths[b], N, 1.f, best + b, nullptr, nt, order,
                                         heads + (size_t)b * N);
    return 0;
}
int emu_dmv1o(const void* dec, const void* attach, const int64_t* lengths, int B, int N, int in_dtype, int semiring,
              const float* glogZ, float* logZ, float* gdec, float* gatt, int nt, int order) {
    return in_dtype == 0 ? dmv_batch<vlg::F32In>(dec, attach, lengths, B, N, semiring, glogZ, logZ, gdec, gatt, nt, order)
                         : dmv_batch<vlg::BF16In>(dec, attach, lengths, B, N, semiring, glogZ, logZ, gdec, gatt, nt, order);
}
int emu_deptree(const void* arc, const int64_t* lengths, int B, int N, int in_dtype, int semiring, const float* glogZ,
                float* logZ, float* garc, int nt, int order) {
    return in_dtype == 0 ? dep_batch<vlg::F32In>(arc, lengths, B, N, semiring, glogZ, logZ, garc, nt, order)
                         : dep_batch<vlg::BF16In>(arc, lengths, B, N, semiring, glogZ, logZ, garc, nt, order);
}
}
