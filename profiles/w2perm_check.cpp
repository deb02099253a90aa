// Torch-free check of the side libraries (librrl_hip_w2perm*.so) against the default library THROUGH THE C ABI:
// both are dlopen'ed in one process, fed the same device buffers, and every output is compared byte for byte; then the
// launches are timed with HIP events.  Starts in milliseconds (no Python, no torch import): written for the last GPU minutes
// of round 3.      hipcc -O2 -o profiles/_ab_w2perm_check profiles/w2perm_check.cpp -ldl
//                  profiles/_ab_w2perm_check recovery_rl_amd/csrc [side library file names ...]
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

typedef int (*fwd_t)(int, int, int, int, int, const float*, int, const float*, const float*, const float*, const float*,
                     const float*, const float*, float*, float*, float*, float*, int, void*);
typedef int (*hid_t)(int, int, int, const float*, const float*, const float*, float*, float*, float*, void*);
typedef int (*gemm_t)(int, int, int, int, int, const float*, int, long long, const float*, int, long long, float*, int,
                      long long, const float*, long long, int, const float*, int, long long, float*, long long, int, void*);

struct Lib {
    std::string name;
    fwd_t fwd;
    hid_t hid;
    gemm_t gemm;
};

#define HIP(x)                                                                          \
    do {                                                                                \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess) {                                                         \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                                    \
        }                                                                               \
    } while (0)

static uint64_t lcg = 88172645463325252ULL;
static float rnd() {      // [-1, 1)
    lcg = lcg * 6364136223846793005ULL + 1442695040888963407ULL;
    return float(int32_t(lcg >> 33) - (1 << 30)) / float(1 << 30);
}
struct Buf {
    float* d = nullptr;
    size_t n = 0;
    explicit Buf(size_t n_, float scale = 0.f, bool relu = false) : n(n_) {
        HIP(hipMalloc(&d, n * sizeof(float)));
        std::vector<float> h(n, 0.f);
        if (scale != 0.f)
            for (auto& v : h) {
                v = rnd() * scale;
                if (relu && v < 0.f) v = 0.f;
            }
        HIP(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
    }
    void clear() { HIP(hipMemset(d, 0xff, n * sizeof(float))); }      // NaN pattern: an element nobody wrote shows
    std::vector<float> host() const {
        std::vector<float> h(n);
        HIP(hipMemcpy(h.data(), d, n * sizeof(float), hipMemcpyDeviceToHost));
        return h;
    }
    ~Buf() { (void)hipFree(d); }
};

static bool load(const std::string& dir, const std::string& file, Lib& lib) {
    const std::string path = dir + "/" + file;
    void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        printf("cannot load %s: %s\n", path.c_str(), dlerror());
        return false;
    }
    lib.name = file;
    lib.fwd = (fwd_t)dlsym(h, "rrl_mlp3_forward");
    lib.hid = (hid_t)dlsym(h, "rrl_mlp_hidden_backward");
    lib.gemm = (gemm_t)dlsym(h, "rrl_gemm_f32");
    return lib.fwd && lib.hid && lib.gemm;
}

typedef std::vector<std::vector<float>> Result;
static int compare(const char* what, const Lib& lib, const Result& want, const Result& got) {
    int bad = 0;
    for (size_t k = 0; k < want.size(); ++k)
        if (want[k].size() != got[k].size() || memcmp(want[k].data(), got[k].data(), want[k].size() * sizeof(float))) {
            size_t diff = 0, first = 0;
            for (size_t e = 0; e < want[k].size(); ++e)
                if (memcmp(&want[k][e], &got[k][e], 4)) {
                    if (!diff) first = e;
                    ++diff;
                }
            printf("  DIFFERENT %-28s %-34s output %zu: %zu of %zu elements, first at %zu (%g vs %g)\n", lib.name.c_str(), what, k,
                   diff, want[k].size(), first, want[k][first], got[k][first]);
            ++bad;
        }
    if (!bad) printf("  identical %-28s %s\n", lib.name.c_str(), what);
    return bad;
}

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : "recovery_rl_amd/csrc";
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        printf("no GPU\n");
        return 3;
    }
    hipDeviceProp_t prop;
    HIP(hipGetDeviceProperties(&prop, 0));
    printf("device: %s (%s), %d CUs\n", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    std::vector<Lib> libs;
    std::vector<std::string> files = {"librrl_hip.so", "librrl_hip_w2perm.so", "librrl_hip_w2perm_all.so", "librrl_hip_w2perm_bwd.so"};
    if (argc > 2) {      // explicit list of side libraries (file names inside <dir>); the default library is always first
        files.resize(1);
        for (int k = 2; k < argc; ++k) files.push_back(argv[k]);
    }
    for (const auto& f : files) {
        Lib l;
        if (load(dir, f, l)) libs.push_back(l);
    }
    if (libs.size() < 2) return 4;
    const int H = 256;
    int bad = 0;
    hipStream_t st;
    HIP(hipStreamCreate(&st));

    // ---- fused stack forward, split path (scratch given) --------------------------------------------------------------
    const int shapes[][4] = {{4096, 4, 1, 2}, {4096, 2, 4, 1}, {2061, 4, 1, 2}, {1040, 2, 4, 1}, {1024, 4, 1, 2}, {256, 4, 1, 2},
                             {256, 2, 4, 1}, {100, 4, 1, 2}};
    for (const auto& s : shapes) {
        const int M = s[0], din = s[1], dout = s[2], G = s[3];
        Buf x(size_t(M) * din, 3.f), W1(size_t(G) * H * din, 1.f), b1(size_t(G) * H, 1.f), W2(size_t(G) * H * H, 1.f / 16),
            b2(size_t(G) * H, 1.f), W3(size_t(G) * dout * H, 1.f / 16), b3(size_t(G) * dout, 1.f);
        Buf h1(size_t(G) * M * H), h2(size_t(G) * M * H), out(size_t(G) * M * dout), scratch(size_t(4) * G * M * dout);
        Result want;
        char what[96];
        snprintf(what, sizeof what, "forward M=%d din=%d dout=%d G=%d", M, din, dout, G);
        for (size_t k = 0; k < libs.size(); ++k) {
            h1.clear(), h2.clear(), out.clear(), scratch.clear();
            const int rc = libs[k].fwd(G, M, H, din, dout, x.d, din, W1.d, b1.d, W2.d, b2.d, W3.d, b3.d, h1.d, h2.d, out.d,
                                       scratch.d, 1, st);
            HIP(hipStreamSynchronize(st));
            if (rc) printf("  %s: rc %d\n", libs[k].name.c_str(), rc), ++bad;
            Result got = {h1.host(), h2.host(), out.host()};
            if (k == 0) {
                want = got;
                size_t nan = 0;
                for (const auto& v : want)
                    for (float e : v) nan += e != e;
                printf("%s: default library wrote %zu NaN (expected 0), out[0] = %g\n", what, nan, want[2][0]);
                bad += nan != 0;
            } else {
                bad += compare(what, libs[k], want, got);
            }
        }
    }
    // ---- hidden layer of the backward (tile form) and the NT GEMM: load_direct operands ---------------------------------
    {
        const int G = 2, B = 256;
        Buf dh2(size_t(G) * B * H, 1.f), h1(size_t(G) * B * H, 1.f, true), W2(size_t(G) * H * H, 1.f / 16);
        Buf dW2(size_t(G) * H * H), db2(size_t(G) * H), dh1(size_t(G) * B * H);
        Buf A(size_t(G) * B * H, 1.f), Bm(size_t(G) * H * H, 1.f / 16), bias(size_t(G) * H, 1.f), C(size_t(G) * B * H);
        Result want_h, want_g;
        for (size_t k = 0; k < libs.size(); ++k) {
            dW2.clear(), db2.clear(), dh1.clear(), C.clear();
            int rc = libs[k].hid(G, B, H, dh2.d, h1.d, W2.d, dW2.d, db2.d, dh1.d, st);
            rc |= libs[k].gemm(0, G, B, H, H, A.d, H, (long long)B * H, Bm.d, H, (long long)H * H, C.d, H, (long long)B * H, bias.d, H,
                               1, nullptr, 0, 0, nullptr, 0, 0, st);
            HIP(hipStreamSynchronize(st));
            if (rc) printf("  %s: rc %d\n", libs[k].name.c_str(), rc), ++bad;
            Result got_h = {dW2.host(), db2.host(), dh1.host()}, got_g = {C.host()};
            if (k == 0) {
                want_h = got_h, want_g = got_g;
                size_t nan = 0;
                for (const auto& v : {want_h[0], want_h[1], want_h[2], want_g[0]})
                    for (float e : v) nan += e != e;
                printf("hidden backward + NT GEMM (G=2, B=256, H=256): default library wrote %zu NaN (expected 0)\n", nan);
                bad += nan != 0;
            } else {
                bad += compare("hidden backward G=2 B=256", libs[k], want_h, got_h);
                bad += compare("gemm NT bias relu 256^3 G=2", libs[k], want_g, got_g);
            }
        }
    }
    // ---- timing: back-to-back launches on one stream, HIP events; libraries interleaved, three rounds, best of ----------
    {
        hipEvent_t e0, e1;
        HIP(hipEventCreate(&e0));
        HIP(hipEventCreate(&e1));
        const int reps = 300;
        struct Case { const char* name; int M, din, dout, G; } cases[] = {{"forward 4096 rows, 2 heads (Q / Q_risk acting)", 4096, 4, 1, 2},
                                                                        {"forward 4096 rows, 1 head  (policy acting)", 4096, 2, 4, 1},
                                                                        {"forward  256 rows, 2 heads (update batch)", 256, 4, 1, 2}};
        for (const auto& c : cases) {
            Buf x(size_t(c.M) * c.din, 3.f), W1(size_t(c.G) * H * c.din, 1.f), b1(size_t(c.G) * H, 1.f),
                W2(size_t(c.G) * H * H, 1.f / 16), b2(size_t(c.G) * H, 1.f), W3(size_t(c.G) * c.dout * H, 1.f / 16),
                b3(size_t(c.G) * c.dout, 1.f), out(size_t(c.G) * c.M * c.dout), scratch(size_t(4) * c.G * c.M * c.dout);
            std::vector<float> best(libs.size(), 1e30f);
            for (int round = 0; round < 3; ++round)
                for (size_t k = 0; k < libs.size(); ++k) {
                    auto run = [&] {
                        libs[k].fwd(c.G, c.M, H, c.din, c.dout, x.d, c.din, W1.d, b1.d, W2.d, b2.d, W3.d, b3.d, nullptr, nullptr,
                                    out.d, scratch.d, 1, st);
                    };
                    for (int i = 0; i < 20; ++i) run();
                    HIP(hipEventRecord(e0, st));
                    for (int i = 0; i < reps; ++i) run();
                    HIP(hipEventRecord(e1, st));
                    HIP(hipEventSynchronize(e1));
                    float ms = 0.f;
                    HIP(hipEventElapsedTime(&ms, e0, e1));
                    if (ms * 1e3f / reps < best[k]) best[k] = ms * 1e3f / reps;
                }
            printf("%s, us per forward (stack kernel + sum of partials, eager launches):\n", c.name);
            for (size_t k = 0; k < libs.size(); ++k) printf("  %-28s %8.2f\n", libs[k].name.c_str(), best[k]);
        }
        const int G = 2, B = 256;
        Buf dh2(size_t(G) * B * H, 1.f), h1(size_t(G) * B * H, 1.f, true), W2(size_t(G) * H * H, 1.f / 16);
        Buf dW2(size_t(G) * H * H), db2(size_t(G) * H), dh1(size_t(G) * B * H);
        std::vector<float> best(libs.size(), 1e30f);
        for (int round = 0; round < 3; ++round)
            for (size_t k = 0; k < libs.size(); ++k) {
                auto run = [&] { libs[k].hid(G, B, H, dh2.d, h1.d, W2.d, dW2.d, db2.d, dh1.d, st); };
                for (int i = 0; i < 20; ++i) run();
                HIP(hipEventRecord(e0, st));
                for (int i = 0; i < reps; ++i) run();
                HIP(hipEventRecord(e1, st));
                HIP(hipEventSynchronize(e1));
                float ms = 0.f;
                HIP(hipEventElapsedTime(&ms, e0, e1));
                if (ms * 1e3f / reps < best[k]) best[k] = ms * 1e3f / reps;
            }
        printf("hidden-layer backward G=2 B=256 H=256, us per launch (eager launches):\n");
        for (size_t k = 0; k < libs.size(); ++k) printf("  %-28s %8.2f\n", libs[k].name.c_str(), best[k]);
    }
    printf(bad ? "RESULT: %d comparisons DIFFER\n" : "RESULT: every variant equals the default library bit for bit\n", bad);
    return bad ? 1 : 0;
}
