// timing.hip — per-kernel-family HIP-event timing for bench.py's roofline leg, and the version string.
#include "common.h"
#include <mutex>
#include <vector>

namespace {
struct Rec { hipEvent_t a, b; int fam; };
std::mutex g_mu;
int g_on = 0;
int g_mask = ~0;             // families that are timed when timing is on
std::vector<Rec> g_pool;          // created lazily, reused
size_t g_used = 0;
double g_ms[VARHIP_NFAM], g_flops[VARHIP_NFAM], g_bytes[VARHIP_NFAM];
int64_t g_n[VARHIP_NFAM];
const size_t kMaxRecs = 1 << 15;

void drain_locked() {
    for (size_t i = 0; i < g_used; ++i) {
        float ms = 0.f;
        if (hipEventSynchronize(g_pool[i].b) == hipSuccess && hipEventElapsedTime(&ms, g_pool[i].a, g_pool[i].b) == hipSuccess)
            g_ms[g_pool[i].fam] += ms;
    }
    g_used = 0;
}
}  // namespace

int vh_timing_on(int fam) { return g_on && ((g_mask >> fam) & 1); }

void vh_timing_begin(int fam, hipStream_t s, double flops, double bytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_used == kMaxRecs) drain_locked();
    if (g_used == g_pool.size()) {
        Rec r; r.fam = fam;
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
        g_pool.push_back(r);
    }
    g_pool[g_used].fam = fam;
    (void)hipEventRecord(g_pool[g_used].a, s);
    g_flops[fam] += flops; g_bytes[fam] += bytes; g_n[fam] += 1;
}

void vh_timing_end(int fam, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    (void)fam;
    if (g_used < g_pool.size()) { (void)hipEventRecord(g_pool[g_used].b, s); ++g_used; }
}

// switches shared by the fp16 and bf16 builds of the 16-bit kernels (elem16.h); the environment variables are for experiments
int vh_g_force_tile16 = -1;
int vh_g_gemm16_persist = [] { const char* e = getenv("VARHIP_GEMM16_PERSIST"); return e ? (atoi(e) != 0) : 1; }();
int vh_g_gemm16_deep = [] { const char* e = getenv("VARHIP_GEMM16_DEEP"); return e ? (atoi(e) != 0) : 1; }();       // 0: the 2-stage 64-row tiles only
int vh_g_conv16_force_wm = 0;

extern "C" {

// testing / experiments: force the tile of the next 16-bit GEMM calls (0: 128x128, 1: 64x64 and smaller (64x128 for q/k/v), 2: 256x256, 3: 192x256, -1: automatic)
int varhip_gemm16_force_tile(int tile) { vh_g_force_tile16 = (tile >= 0 && tile <= 3) ? tile : -1; return 0; }
// 0 = whole 256x256 tiles on k_gemm16<8,4,2,4> (one workgroup per tile), 1 (default) = on the persistent k_gemm16p
int varhip_gemm16_persistent(int on) { vh_g_gemm16_persist = on ? 1 : 0; return 0; }
// 0: by size; 2 / 4 / 8: force the 128-pixel / 256-pixel / halo-patch conv kernel (tests, tools/bench_kernels.py)
int varhip_conv16_force_tile(int wm) { vh_g_conv16_force_wm = (wm == 2 || wm == 4 || wm == 8) ? wm : 0; return 0; }

const char* varhip_version(void) { return "var_hip 0.1.0 gfx950"; }

int varhip_timing_enable(int on) { std::lock_guard<std::mutex> lk(g_mu); g_on = on ? 1 : 0; return 0; }
int varhip_timing_select(int family_mask) { std::lock_guard<std::mutex> lk(g_mu); g_mask = family_mask; return 0; }

int varhip_timing_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    drain_locked();
    for (int i = 0; i < VARHIP_NFAM; ++i) { g_ms[i] = g_flops[i] = g_bytes[i] = 0.0; g_n[i] = 0; }
    return 0;
}

int varhip_timing_read(double* ms, double* flops, double* bytes, int64_t* launches) {
    std::lock_guard<std::mutex> lk(g_mu);
    drain_locked();
    for (int i = 0; i < VARHIP_NFAM; ++i) {
        if (ms) ms[i] = g_ms[i];
        if (flops) flops[i] = g_flops[i];
        if (bytes) bytes[i] = g_bytes[i];
        if (launches) launches[i] = g_n[i];
    }
    return VARHIP_NFAM;
}

const char* varhip_timing_name(int f) {
    static const char* names[VARHIP_NFAM] = {"gemm", "conv3x3", "attn", "sampler", "ln", "qkv_prep", "gn", "other", "gemm_small", "conv_small",
                                               "gemm16", "gemm16_small", "conv16h", "conv16_small", "attn16"};
    return (f >= 0 && f < VARHIP_NFAM) ? names[f] : "?";
}

}  // extern "C"
