// Torch-free timing of the PACKED forward launches (rrl_mlp3_forward_multi_packed) through the C ABI at S = 1 .. 16 seeds, for the
// in-tree library and for side builds (timing ablations: profiles/patches/round5_fwd_packed_ablate.patch) -- the launches that
// separate the packed legs of the bench from their targets:
//   A  acting pass, launch 1: task policy + recovery policy on the seed's 4096 observations (2 stacks, G = 1, din 2, dout 4 / 2)
//   B  acting pass, launch 2: twin Q_risk on [s | a] with the task policy's head as input head (G = 2, din 4, dout 1)
//   C  an update's forward:   twin critic (G = 2, din 4, dout 1) + policy (din 2, dout 4) on the seed's 256-row batch, h1 / h2 kept
// Every seed has its own weights and buffers.  Outputs of a side library are compared with the first library's bit for bit.
//     hipcc -O2 -o profiles/_ab_fwd_packed_probe profiles/fwd_packed_probe.cpp -ldl
//     profiles/_ab_fwd_packed_probe recovery_rl_amd/csrc/librrl_hip.so [side.so | NAME=VALUE@copy_of_library.so ...]
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../include/rrl_hip.h"

typedef int (*packed_t)(int, const int*, const rrl_stack_t* const*, void*);

#define HIP(x)                                                                            \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                                      \
        }                                                                                 \
    } while (0)

static uint64_t lcg = 88172645463325252ULL;
static float rnd() {
    lcg = lcg * 6364136223846793005ULL + 1442695040888963407ULL;
    return float(int32_t(lcg >> 33) - (1 << 30)) / float(1 << 30);
}
struct Buf {
    float* d = nullptr;
    size_t n = 0;
    explicit Buf(size_t n_, float scale = 0.f) : n(n_) {
        HIP(hipMalloc(&d, n * sizeof(float)));
        std::vector<float> h(n, 0.f);
        if (scale != 0.f)
            for (auto& v : h) v = rnd() * scale;
        HIP(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
    }
    std::vector<float> host() const {
        std::vector<float> h(n);
        HIP(hipMemcpy(h.data(), d, n * sizeof(float), hipMemcpyDeviceToHost));
        return h;
    }
};
typedef int (*pack_t)(int, int, const float*, float*, void*);
static pack_t g_pack = nullptr;          // rrl_w2_pack of the first library (PROBE_ROW_MAJOR=1: W2 row-major only)
struct Net {
    Buf W1, b1, W2, b2, W3, b3, W2p;
    int G, din, dout;
    Net(int G_, int din_, int dout_)
        : W1(size_t(G_) * 256 * din_, 0.5f), b1(size_t(G_) * 256, 0.2f), W2(size_t(G_) * 256 * 256, 1.f / 16), b2(size_t(G_) * 256, 0.2f),
          W3(size_t(G_) * dout_ * 256, 1.f / 16), b3(size_t(G_) * dout_, 0.2f), W2p(size_t(G_) * 256 * 256), G(G_), din(din_), dout(dout_) {
        if (g_pack && g_pack(G, 256, W2.d, W2p.d, nullptr) != 0) { printf("rrl_w2_pack failed\n"); exit(3); }
        HIP(hipDeviceSynchronize());
    }
};
static rrl_stack_t stack(const Net& n, int M, const float* x, int ldx, float* out, float* scratch, float* h1 = nullptr, float* h2 = nullptr) {
    rrl_stack_t s;
    memset(&s, 0, sizeof s);
    s.G = n.G; s.M = M; s.H = 256; s.din = n.din; s.dout = n.dout; s.ldx = ldx; s.x = x;
    s.W1 = n.W1.d; s.b1 = n.b1.d; s.W2 = n.W2.d; s.b2 = n.b2.d; s.W3 = n.W3.d; s.b3 = n.b3.d;
    s.h1 = h1; s.h2 = h2; s.out = out; s.scratch = scratch;
    s.W2p = g_pack ? n.W2p.d : nullptr;
    return s;
}
// one seed's networks and buffers
struct Seed {
    static constexpr int M = 4096, B = 256;
    Net pol, rec, qr, crit;
    Buf obs, xa, eps, scale, bias, o_pol, s_pol, o_rec, s_rec, o_qr, s_qr, logp;
    Buf xb, ob, o_c, s_c, h1c, h2c, o_p, s_p, h1p, h2p;
    rrl_stack_t A[2], Bq[1], C[2];
    Seed()
        : pol(1, 2, 4), rec(1, 2, 2), qr(2, 4, 1), crit(2, 4, 1), obs(size_t(M) * 2, 20.f), xa(size_t(M) * 4, 1.f), eps(size_t(M) * 2, 1.f),
          scale(2, 0.f), bias(2, 0.f), o_pol(size_t(M) * 4), s_pol(size_t(4) * M * 4), o_rec(size_t(M) * 2), s_rec(size_t(4) * M * 2),
          o_qr(size_t(2) * M), s_qr(size_t(4) * 2 * M), logp(M), xb(size_t(B) * 4, 1.f), ob(size_t(B) * 2, 20.f), o_c(size_t(2) * B),
          s_c(size_t(4) * 2 * B), h1c(size_t(2) * B * 256), h2c(size_t(2) * B * 256), o_p(size_t(B) * 4), s_p(size_t(4) * B * 4),
          h1p(size_t(B) * 256), h2p(size_t(B) * 256) {
        const float one[2] = {1.f, 1.f};
        HIP(hipMemcpy(scale.d, one, 8, hipMemcpyHostToDevice));
        A[0] = stack(pol, M, obs.d, 2, o_pol.d, s_pol.d);
        A[1] = stack(rec, M, obs.d, 2, o_rec.d, s_rec.d);
        Bq[0] = stack(qr, M, xa.d, 4, o_qr.d, s_qr.d);
        rrl_stack_t& l = Bq[0];
        l.use_in_head = 1;
        l.in_head.kind = RRL_HEAD_GAUSS; l.in_head.B = M; l.in_head.head = s_pol.d; l.in_head.n_part = 4;
        l.in_head.part_stride = (long long)M * 4; l.in_head.eps = eps.d; l.in_head.scale = scale.d; l.in_head.bias = bias.d;
        l.in_head.action = xa.d + 2; l.in_head.ld_action = 4; l.in_head.logp = logp.d; l.in_head.obs_in = obs.d;
        l.in_head.obs_out = xa.d;
        C[0] = stack(crit, B, xb.d, 4, o_c.d, s_c.d, h1c.d, h2c.d);
        C[1] = stack(pol, B, ob.d, 2, o_p.d, s_p.d, h1p.d, h2p.d);
    }
    std::vector<std::vector<float>> outputs() const {
        return {s_pol.host(), s_rec.host(), s_qr.host(), xa.host(), logp.host(), s_c.host(), s_p.host(), h2c.host(), h1p.host()};
    }
};

int main(int argc, char** argv) {
    if (argc < 2) return 1;
    const int reps = getenv("PROBE_REPS") ? atoi(getenv("PROBE_REPS")) : 200, kMax = 16;   // PROBE_S=<S>: that seed count only
    hipStream_t st;
    HIP(hipStreamCreate(&st));
    if (!getenv("PROBE_ROW_MAJOR")) {
        void* h0 = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
        g_pack = h0 ? (pack_t)dlsym(h0, "rrl_w2_pack") : nullptr;
    }
    printf("W2 in fragment order: %s\n", g_pack ? "yes" : "no (row-major loads)");
    std::vector<std::unique_ptr<Seed>> seeds;
    for (int s = 0; s < kMax; ++s) seeds.emplace_back(new Seed());
    hipEvent_t e0, e1;
    HIP(hipEventCreate(&e0));
    HIP(hipEventCreate(&e1));
    std::vector<std::vector<float>> want;
    const int counts[] = {1, 2, 4, 8, 16};
    for (int k = 1; k < argc; ++k) {
        // "NAME=VALUE@library.so": the switch is set while the library is loaded and called for the first time (the libraries
        // read their switches once); give such a library its own copy of the file, or dlopen returns the handle it already has
        std::string arg = argv[k], env;
        const size_t at = arg.find('@');
        if (at != std::string::npos) {
            env = arg.substr(0, at);
            arg = arg.substr(at + 1);
            const size_t eq = env.find('=');
            setenv(env.substr(0, eq).c_str(), env.substr(eq + 1).c_str(), 1);
        }
        void* h = dlopen(arg.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!h) { printf("dlopen: %s\n", dlerror()); return 4; }
        packed_t fn = (packed_t)dlsym(h, "rrl_mlp3_forward_multi_packed");
        if (!fn) { printf("%s: no packed entry\n", argv[k]); continue; }
        printf("%s\n", argv[k]);
        for (int S : counts) {
            if (getenv("PROBE_S") && atoi(getenv("PROBE_S")) != S) continue;
            int nA[kMax], nB[kMax], nC[kMax];
            const rrl_stack_t *mA[kMax], *mB[kMax], *mC[kMax];
            for (int s = 0; s < S; ++s) {
                nA[s] = 2; nB[s] = 1; nC[s] = 2;
                mA[s] = seeds[s]->A; mB[s] = seeds[s]->Bq; mC[s] = seeds[s]->C;
            }
            int rc = fn(S, nA, mA, st);
            rc |= fn(S, nB, mB, st);
            rc |= fn(S, nC, mC, st);
            HIP(hipStreamSynchronize(st));
            if (rc) { printf("  S = %d: rc %d\n", S, rc); continue; }
            if (!env.empty() && S == 16) unsetenv(env.substr(0, env.find('=')).c_str());
            const char* verdict = "";
            if (S == 16 && !getenv("PROBE_S")) {
                std::vector<std::vector<float>> got;
                for (int s = 0; s < S; s += 5)
                    for (auto& v : seeds[s]->outputs()) got.push_back(v);
                if (k == 1) want = got, verdict = "(reference)";
                else {
                    int same = 1;
                    for (size_t j = 0; j < got.size(); ++j) same &= memcmp(want[j].data(), got[j].data(), got[j].size() * 4) == 0;
                    verdict = same ? "outputs identical" : "outputs DIFFERENT";
                }
            }
            float t[3];
            for (int which = 0; which < 3; ++which) {
                const int* n = which == 0 ? nA : (which == 1 ? nB : nC);
                const rrl_stack_t* const* m = which == 0 ? mA : (which == 1 ? mB : mC);
                for (int w = 0; w < 10; ++w) fn(S, n, m, st);
                HIP(hipEventRecord(e0, st));
                for (int r = 0; r < reps; ++r) fn(S, n, m, st);
                HIP(hipEventRecord(e1, st));
                HIP(hipEventSynchronize(e1));
                HIP(hipEventElapsedTime(&t[which], e0, e1));
                t[which] *= 1000.f / reps;
            }
            // f32 MFMA work of layer 2: 2 * 256 * 256 flop per row and head
            const double fl = 2.0 * 256 * 256 * S;
            printf("  S = %2d   policies %7.2f us (%.2f of peak)   Q_risk + head %7.2f us (%.2f)   update forward %7.2f us (%.2f)   %s\n", S,
                   t[0], fl * 2 * 4096 / (t[0] * 1e-6) / 157.3e12, t[1], fl * 2 * 4096 / (t[1] * 1e-6) / 157.3e12, t[2],
                   fl * 3 * 256 / (t[2] * 1e-6) / 157.3e12, verdict);
        }
    }
    return 0;
}
