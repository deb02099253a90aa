// Torch-free timing of the acting pass's two forward launches (rrl_mlp3_forward_multi) through the C ABI, for the in-tree
// library and for side builds (timing ablations of the stream form: profiles/patches/stream_ablate.patch):
//   launch 1: task policy + recovery policy on the same observations (2 stacks, G = 1, din 2, dout 4 / 2)
//   launch 2: twin Q_risk on [s | a] (1 stack, G = 2, din 4, dout 1), optionally with the task policy's head as input head
// Every library is timed with HIP events over a run of back-to-back launches; the column-split kernel is the same library
// with RRL_FWD_STREAM=0 (a second copy of the .so, so that its switch is read on its own) and its outputs are the reference
// the stream form must equal bit for bit.
//     hipcc -O2 -o profiles/_ab_fwd_stream_probe profiles/fwd_stream_probe.cpp -ldl
//     profiles/_ab_fwd_stream_probe recovery_rl_amd/csrc/librrl_hip.so [side.so ...]
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/rrl_hip.h"

typedef int (*multi_t)(int, const rrl_stack_t*, void*);

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
struct Net {
    Buf W1, b1, W2, b2, W3, b3;
    int G, din, dout;
    Net(int G_, int din_, int dout_)
        : W1(size_t(G_) * 256 * din_, 0.5f), b1(size_t(G_) * 256, 0.2f), W2(size_t(G_) * 256 * 256, 1.f / 16), b2(size_t(G_) * 256, 0.2f),
          W3(size_t(G_) * dout_ * 256, 1.f / 16), b3(size_t(G_) * dout_, 0.2f), G(G_), din(din_), dout(dout_) {}
};
static rrl_stack_t stack(const Net& n, int M, const float* x, int ldx, float* out, float* scratch) {
    rrl_stack_t s;
    memset(&s, 0, sizeof s);
    s.G = n.G; s.M = M; s.H = 256; s.din = n.din; s.dout = n.dout; s.ldx = ldx; s.x = x;
    s.W1 = n.W1.d; s.b1 = n.b1.d; s.W2 = n.W2.d; s.b2 = n.b2.d; s.W3 = n.W3.d; s.b3 = n.b3.d;
    s.h1 = nullptr; s.h2 = nullptr; s.out = out; s.scratch = scratch;
    return s;
}

int main(int argc, char** argv) {
    if (argc < 2) return 1;
    const int M = getenv("PROBE_M") ? atoi(getenv("PROBE_M")) : 4096;
    const int reps = 300;
    hipStream_t st;
    HIP(hipStreamCreate(&st));
    Net pol(1, 2, 4), rec(1, 2, 2), qr(2, 4, 1);
    Buf obs(size_t(M) * 2, 20.f), xa(size_t(M) * 4, 1.f), eps(size_t(M) * 2, 1.f), scale(2, 0.f), bias(2, 0.f);
    {
        const float one[2] = {1.f, 1.f};
        HIP(hipMemcpy(scale.d, one, 8, hipMemcpyHostToDevice));
    }
    Buf o_pol(size_t(M) * 4), s_pol(size_t(4) * M * 4), o_rec(size_t(M) * 2), s_rec(size_t(4) * M * 2), o_qr(size_t(2) * M), s_qr(size_t(4) * 2 * M);
    Buf logp(M);

    // the first library also runs with RRL_FWD_STREAM=0 (through a copy of the file): the column-split reference
    std::vector<std::string> names;
    std::vector<multi_t> fns;
    {
        const std::string copy = "/tmp/_probe_split_copy.so";
        std::string cmd = std::string("cp ") + argv[1] + " " + copy;
        if (system(cmd.c_str()) != 0) return 5;
        setenv("RRL_FWD_STREAM", "0", 1);
        void* h = dlopen(copy.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!h) { printf("dlopen: %s\n", dlerror()); return 4; }
        multi_t f = (multi_t)dlsym(h, "rrl_mlp3_forward_multi");
        rrl_stack_t one = stack(pol, M, obs.d, 2, o_pol.d, s_pol.d);
        f(1, &one, st);                           // reads the switch
        HIP(hipStreamSynchronize(st));
        unsetenv("RRL_FWD_STREAM");
        names.push_back("column-split (RRL_FWD_STREAM=0)");
        fns.push_back(f);
    }
    for (int k = 1; k < argc; ++k) {
        void* h = dlopen(argv[k], RTLD_NOW | RTLD_LOCAL);
        if (!h) { printf("dlopen: %s\n", dlerror()); return 4; }
        names.push_back(argv[k]);
        fns.push_back((multi_t)dlsym(h, "rrl_mlp3_forward_multi"));
    }
    hipEvent_t e0, e1;
    HIP(hipEventCreate(&e0));
    HIP(hipEventCreate(&e1));
    std::vector<std::vector<float>> want;
    for (size_t k = 0; k < fns.size(); ++k) {
        rrl_stack_t l1[2] = {stack(pol, M, obs.d, 2, o_pol.d, s_pol.d), stack(rec, M, obs.d, 2, o_rec.d, s_rec.d)};
        rrl_stack_t l2 = stack(qr, M, xa.d, 4, o_qr.d, s_qr.d);
        rrl_stack_t l2h = l2;
        l2h.use_in_head = 1;
        l2h.in_head.kind = RRL_HEAD_GAUSS; l2h.in_head.B = M; l2h.in_head.head = s_pol.d; l2h.in_head.n_part = 4;
        l2h.in_head.part_stride = (long long)M * 4; l2h.in_head.eps = eps.d; l2h.in_head.scale = scale.d; l2h.in_head.bias = bias.d;
        l2h.in_head.action = xa.d + 2; l2h.in_head.ld_action = 4; l2h.in_head.logp = logp.d; l2h.in_head.obs_in = obs.d;
        l2h.in_head.obs_out = xa.d;
        HIP(hipMemset(s_pol.d, 0, s_pol.n * 4)); HIP(hipMemset(s_rec.d, 0, s_rec.n * 4)); HIP(hipMemset(s_qr.d, 0, s_qr.n * 4));
        int rc = fns[k](2, l1, st);
        rc |= fns[k](1, &l2h, st);
        HIP(hipStreamSynchronize(st));
        if (rc) { printf("%s: rc %d\n", names[k].c_str(), rc); continue; }
        std::vector<std::vector<float>> got = {s_pol.host(), s_rec.host(), s_qr.host(), xa.host(), logp.host()};
        int same = 1;
        if (k == 0) want = got;
        else for (size_t j = 0; j < got.size(); ++j) same &= memcmp(want[j].data(), got[j].data(), got[j].size() * 4) == 0;
        float t[3];
        for (int which = 0; which < 3; ++which) {
            for (int w = 0; w < 20; ++w) which == 0 ? fns[k](2, l1, st) : (which == 1 ? fns[k](1, &l2, st) : fns[k](1, &l2h, st));
            HIP(hipEventRecord(e0, st));
            for (int r = 0; r < reps; ++r) which == 0 ? fns[k](2, l1, st) : (which == 1 ? fns[k](1, &l2, st) : fns[k](1, &l2h, st));
            HIP(hipEventRecord(e1, st));
            HIP(hipEventSynchronize(e1));
            HIP(hipEventElapsedTime(&t[which], e0, e1));
            t[which] *= 1000.f / reps;
        }
        printf("%-60s M=%d  policies %6.2f us   Q_risk %6.2f us   Q_risk+head %6.2f us   outputs %s\n", names[k].c_str(), M, t[0], t[1],
               t[2], k == 0 ? "(reference)" : (same ? "identical" : "DIFFERENT"));
    }
    return 0;
}
