// Development lab for the persistent 8-phase GEMM (ecamp_amd/csrc/gemm_q8.h): a torch-free binary that compiles the SAME kernel
// header as the product, checks it against the product's 128^2 kernel (ecamp_gemm with q8_mode = 0) and times variants.
//   make -C tools/gemm_lab        (cross-compiles here)          gpurun -- tools/gemm_lab/lab [shape-substr ...]
#include "gemm_q4.h"
#include "../../ecamp_amd/csrc/gemm_q16.h"
#include "../../include/ecamp_hip.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <string>
#include <algorithm>
#include <math.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

struct Shape { const char* name; int M, N, K; };
static const Shape SHAPES[] = {
    {"enc_qkv", 12800, 2304, 768}, {"enc_fc1", 12800, 3072, 768}, {"enc_fc2", 12800, 768, 3072}, {"enc_proj", 12800, 768, 768},
    {"dec_qkv", 50432, 1536, 512}, {"dec_fc1", 50432, 2048, 512}, {"dec_fc2", 50432, 512, 2048},
    {"bert_qkv", 32768, 2304, 768}, {"bert_dense", 32768, 768, 768}, {"bert_inter", 32768, 1536, 768}, {"bert_out", 32768, 768, 1536},
    {"vocab", 32768, 30000, 768}, {"sq4k", 4096, 4096, 4096}, {"sq8k", 8192, 8192, 8192}, {"ragged", 1000, 520, 200},
    // K sweep on one exact round of 256 tiles (per-tile overhead = intercept), and three exact rounds at the model's K
    {"ks256", 4096, 4096, 256}, {"ks512", 4096, 4096, 512}, {"ks1024", 4096, 4096, 1024}, {"ks2048", 4096, 4096, 2048}, {"r3k768", 12288, 4096, 768},
};

static unsigned short f2bf_h(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
static float bf2f_h(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static float frand() { rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull; return ((rng_state >> 40) & 0xFFFFFF) / 8388608.0f - 1.0f; }

static void fill_bf16(unsigned short* d, size_t n, float scale) {
    std::vector<unsigned short> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = f2bf_h(frand() * scale);
    CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
}

typedef void (*q8_fn)(GemmArgs);
// epi: 0 plain(+bias) 1 bias+pre+gelu 2 +residual 3 gmul 4 f32.  dbg variants only exist for the forward form.
static int g_zero = 0;
static int g_sch = 0;   // schedule variant of the launches that follow (gemm_q8.h SCH)
static q8_fn pick(int a_kc, int b_kc, int epi, int nslot, int dbg) {
    (void)nslot;
#define W(A, B, E, S) ((q8_fn)gemm_bf16_q8_kernel<A, B, E, 0, false, S>)
    if (g_sch == 16 || g_sch == 12) {   // four waves on the 16 x 16 x 32 MFMA (csrc/gemm_q16.h): 256- (16) or 192-column (12) tiles; forward and data-gradient forms, plain / bias / residual
        if (!a_kc) return nullptr;
        if (dbg == 8) {   // data-gradient form with conflict-free (wrong) transpose-read addresses: the price of the 2-way bank meeting
            if (epi != 0 || b_kc) return nullptr;
            return g_sch == 16 ? (q8_fn)gemm_bf16_q16_kernel<0, 8, false, 8> : (q8_fn)gemm_bf16_q16_kernel<0, 6, false, 8>;
        }
        if (dbg != 0) {
            if (g_sch != 16 || epi != 0 || !b_kc) return nullptr;
            if (dbg == 1) return (q8_fn)gemm_bf16_q16_kernel<0, 8, true, 1>;
            if (dbg == 2) return (q8_fn)gemm_bf16_q16_kernel<0, 8, true, 2>;
            if (dbg == 4) return (q8_fn)gemm_bf16_q16_kernel<0, 8, true, 4>;
            return nullptr;
        }
        if (epi != 0 && epi != 2) return nullptr;
        if (b_kc) {
            if (g_sch == 16) return epi == 0 ? (q8_fn)gemm_bf16_q16_kernel<0, 8, true> : (q8_fn)gemm_bf16_q16_kernel<2, 8, true>;
            return epi == 0 ? (q8_fn)gemm_bf16_q16_kernel<0, 6, true> : (q8_fn)gemm_bf16_q16_kernel<2, 6, true>;
        }
        if (g_sch == 16) return epi == 0 ? (q8_fn)gemm_bf16_q16_kernel<0, 8, false> : (q8_fn)gemm_bf16_q16_kernel<2, 8, false>;
        return epi == 0 ? (q8_fn)gemm_bf16_q16_kernel<0, 6, false> : (q8_fn)gemm_bf16_q16_kernel<2, 6, false>;
    }
    if (g_sch == 4 || g_sch == 5) {   // four waves of 128 x 128 (gemm_q4.h): forward form only; 5 = all DMA parts right behind the barrier (EARLY)
        if (!(a_kc && b_kc)) return nullptr;
        if (dbg != 0) {   // timing decomposition of the plain kernel: 1 no MFMA, 2 no DMA, 4 no fragment reads (results are garbage)
            if (g_sch != 4 || epi != 0) return nullptr;
            if (dbg == 1) return (q8_fn)gemm_bf16_q4_kernel<0, false, 1>;
            if (dbg == 2) return (q8_fn)gemm_bf16_q4_kernel<0, false, 2>;
            if (dbg == 4) return (q8_fn)gemm_bf16_q4_kernel<0, false, 4>;
            if (dbg == 6) return (q8_fn)gemm_bf16_q4_kernel<0, false, 6>;
            if (dbg == 5) return (q8_fn)gemm_bf16_q4_kernel<0, false, 5>;
            return nullptr;
        }
        if (g_sch == 4) {
            if (epi == 0) return (q8_fn)gemm_bf16_q4_kernel<0, false>;
            if (epi == 1) return (q8_fn)gemm_bf16_q4_kernel<1, false>;
            if (epi == 2) return (q8_fn)gemm_bf16_q4_kernel<2, false>;
        } else {
            if (epi == 0) return (q8_fn)gemm_bf16_q4_kernel<0, true>;
            if (epi == 1) return (q8_fn)gemm_bf16_q4_kernel<1, true>;
            if (epi == 2) return (q8_fn)gemm_bf16_q4_kernel<2, true>;
        }
        return nullptr;
    }
    if (g_sch == 1) {
        if (dbg == 4 || dbg == 1024) {   // the epilogue's cost split: 4 = no epilogue at all, 1024 = its arithmetic and instructions, every store dropped
            if (!(a_kc && b_kc)) return nullptr;
            if (epi == 0) return dbg == 4 ? (q8_fn)gemm_bf16_q8_kernel<true, true, 0, 4, false, 1> : (q8_fn)gemm_bf16_q8_kernel<true, true, 0, 1024, false, 1>;
            if (epi == 1) return dbg == 4 ? (q8_fn)gemm_bf16_q8_kernel<true, true, 1, 4, false, 1> : (q8_fn)gemm_bf16_q8_kernel<true, true, 1, 1024, false, 1>;
            return nullptr;
        }
        if (dbg != 0) return nullptr;
#define WS(A, B, E) W(A, B, E, 1)
        if (a_kc && b_kc) { if (epi == 0) return WS(true, true, 0); if (epi == 1) return WS(true, true, 1); if (epi == 2) return WS(true, true, 2); }
        if (a_kc && !b_kc) { if (epi == 0) return WS(true, false, 0); if (epi == 3) return WS(true, false, 3); if (epi == 2) return WS(true, false, 2); }
        if (!a_kc && !b_kc && epi == 4) return WS(false, false, 4);
#undef WS
        return nullptr;
    }
#undef W
#define V(A, B, E, D) ((q8_fn)gemm_bf16_q8_kernel<A, B, E, D>)
    if (a_kc && b_kc) {
        if (epi == 1) { if (dbg == 4) return V(true, true, 1, 4); if (dbg == 6) return V(true, true, 1, 6); return V(true, true, 1, 0); }
        if (epi == 0) { if (dbg == 4) return V(true, true, 0, 4); if (dbg == 6) return V(true, true, 0, 6); if (dbg == 5) return V(true, true, 0, 5); return V(true, true, 0, 0); }
        if (epi == 2) return V(true, true, 2, 0);
    }
    if (a_kc && !b_kc) { if (epi == 0) return V(true, false, 0, 0); if (epi == 3) return V(true, false, 3, 0); if (epi == 2) return V(true, false, 2, 0); }
    if (!a_kc && !b_kc && epi == 4) return V(false, false, 4, 0);
#undef V
    return nullptr;
}

struct Run {
    // C[M,N] = opA[M,K] opB[K,N]
    int M, N, K, a_kc, b_kc;
    long lda, ldb, ldc;
    const void *A, *B; void* C;
    const float* bias; void* pre; int act; const void* gmul; const void* residual;
    int out_f32, accumulate, split; float* ws;
};

static void launch_q8(const Run& r, int nslot, int dbg, int grid_override, hipStream_t s) {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = r.A; g.B = r.B; g.C = r.C; g.M = r.M; g.N = r.N; g.K = r.K; g.lda = r.lda; g.ldb = r.ldb; g.ldc = r.ldc;
    g.bias = r.bias; g.pre_out = r.pre; g.ldp = r.N; g.act = r.act; g.gmul = r.gmul; g.ldg = r.N; g.residual = r.residual; g.ldr = r.N;
    g.out_f32 = r.out_f32; g.accumulate = r.accumulate;
    g.alpha = 1.f; g.alpha_out = 1.f;
    int split = r.split < 1 ? 1 : r.split;
    long kps = (r.K + split - 1) / split;
    kps = (kps + 63) / 64 * 64;
    split = (int)((r.K + kps - 1) / kps);
    g.k_per_split = (int)kps; g.nsplit = split;
    g.partial = split > 1 ? r.ws : nullptr;
    g.nbm = (r.M + 255) / 256; g.nbn = (r.N + 255) / 256;
    g.wide = (r.ldc % 8 == 0) && (r.N % 8 == 0);
    g.ldp = g.ldg = g.ldr = r.N;
    g.dbg = dbg;
    const long total = nslot == 12 ? (long)g.nbm * ((r.N + 191) / 192) : (long)g.nbm * g.nbn * split;
    int ncu = grid_override > 0 ? grid_override : 256;
    dim3 grid((unsigned)(total < ncu ? total : ncu));
    const int epi = r.out_f32 ? 4 : r.gmul ? 3 : r.residual ? 2 : r.pre ? 1 : 0;
    g_sch = nslot;
    q8_fn fn = pick(r.a_kc, r.b_kc, epi, nslot, dbg);
    if (!fn) { fprintf(stderr, "no Q8 instance for form %d%d epi %d dbg %d sch %d\n", r.a_kc, r.b_kc, epi, dbg, nslot); return; }
    const size_t shm = (size_t)10 * Q8_HALF;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipLaunchKernelGGL(fn, grid, dim3(nslot == 4 || nslot == 5 || nslot == 16 || nslot == 12 ? 256 : 512), shm, s, g);
}

static int launch_ref(const Run& r, int /*unused*/, hipStream_t s) {   // the product's 128^2 kernel
    ecamp_set_option("q8_mode", 0);
    return ecamp_gemm(r.A, r.B, r.C, r.M, r.N, r.K, r.a_kc, r.lda, r.b_kc, r.ldb, r.ldc, r.bias, r.residual, r.N, r.pre, r.N, r.gmul, r.N, r.act, 1.0f,
                      nullptr, ECAMP_BF16, r.out_f32, r.accumulate, r.split, r.ws, nullptr, (ecampStream_t)s);
}

// order-independent checksum of a device buffer (race screen: a deterministic kernel must reproduce it bit for bit)
__global__ void checksum_kernel(const unsigned* __restrict__ x, size_t n, unsigned long long* out) {
    unsigned long long a = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a += (unsigned long long)x[i] * (2 * (i & 1023) + 1);
    atomicAdd(out, a);
}
static unsigned long long checksum(const void* p, size_t bytes, hipStream_t s) {
    static unsigned long long* d = nullptr;
    if (!d) CK(hipMalloc(&d, 8));
    CK(hipMemsetAsync(d, 0, 8, s));
    hipLaunchKernelGGL(checksum_kernel, dim3(1024), dim3(256), 0, s, (const unsigned*)p, bytes / 4, d);
    unsigned long long h = 0;
    CK(hipMemcpyAsync(&h, d, 8, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    return h;
}

static int g_n = 20;   // launches per timing (--n=; a long run lets tools/power_probe-style sampling see the loop)
template <typename F> static float time_us(F f, int n = 0) {
    if (n == 0) n = g_n;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < n; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return ms * 1000.f / n;
}

static int g_verbose = 0; static size_t g_ld = 1;
static double compare(const void* x, const void* y, size_t n, bool f32, double* ref_norm) {
    std::vector<unsigned char> hx(n * (f32 ? 4 : 2)), hy(hx.size());
    CK(hipMemcpy(hx.data(), x, hx.size(), hipMemcpyDeviceToHost));
    CK(hipMemcpy(hy.data(), y, hy.size(), hipMemcpyDeviceToHost));
    double md = 0, mr = 0;
    for (size_t i = 0; i < n; ++i) {
        float a = f32 ? ((float*)hx.data())[i] : bf2f_h(((unsigned short*)hx.data())[i]);
        float b = f32 ? ((float*)hy.data())[i] : bf2f_h(((unsigned short*)hy.data())[i]);
        if (g_verbose > 0 && !(fabs(a - b) <= 1e-2 * (1 + fabs(b)))) { printf("   mismatch at %zu (row %zu col %zu): got %g want %g\n", i, i / g_ld, i % g_ld, a, b); --g_verbose; }
        if (!(fabs(a - b) <= md)) md = fabs(a - b);   // NaN-propagating
        if (fabs(b) > mr) mr = fabs(b);
    }
    *ref_norm = mr;
    return md;
}

int main(int argc, char** argv) {
    std::vector<std::string> want;
    int forms = 7;        // bit0 fwd, bit1 dgrad, bit2 wgrad
    bool quick = false;
    std::vector<int> dbgs = {0};
    std::vector<int> nslots = {0};   // the list of schedule variants to run (--sch=0,1,2); the name is historical
    int grid_override = 0, stress = 0, hog = 0, plain_only = 0;   // --hog: a small spinning kernel on a second stream beside every launch (uneven load)
    for (int i = 1; i < argc; ++i) {
        if (!strncmp(argv[i], "--stress=", 9)) { stress = atoi(argv[i] + 9); continue; }
        if (!strcmp(argv[i], "--hog")) { hog = 1; continue; }
        if (!strcmp(argv[i], "--plain-only")) { plain_only = 1; continue; }   // forward form: only the plain (no bias / activation) variant
        if (!strncmp(argv[i], "--n=", 4)) { g_n = atoi(argv[i] + 4); continue; }
        if (!strcmp(argv[i], "--zero")) { g_zero = 1; continue; }   // all-zero operands: the matrix pipes draw far less power (is a gap power or structure?)
        if (!strncmp(argv[i], "--forms=", 8)) forms = atoi(argv[i] + 8);
        else if (!strcmp(argv[i], "--quick")) quick = true;
        else if (!strncmp(argv[i], "--grid=", 7)) grid_override = atoi(argv[i] + 7);
        else if (!strncmp(argv[i], "--dbg=", 6)) { dbgs.clear(); char* p = argv[i] + 6; while (*p) { dbgs.push_back((int)strtol(p, &p, 10)); if (*p == ',') ++p; } }
        else if (!strncmp(argv[i], "--sch=", 6)) { nslots.clear(); char* p = argv[i] + 6; while (*p) { nslots.push_back((int)strtol(p, &p, 10)); if (*p == ',') ++p; } }
        else if (!strncmp(argv[i], "--nslot=", 8)) { nslots.clear(); char* p = argv[i] + 8; while (*p) { nslots.push_back((int)strtol(p, &p, 10)); if (*p == ',') ++p; } }
        else want.push_back(argv[i]);
    }
    hipStream_t s; CK(hipStreamCreate(&s));
    printf("%-11s %-5s %6s %6s %5s | %-22s %8s %7s  %s\n", "shape", "form", "M", "N", "K", "variant", "us", "TF", "check");
    for (const Shape& sh : SHAPES) {
        bool sel = want.empty() ? (strcmp(sh.name, "sq8k") && strcmp(sh.name, "vocab")) : false;
        for (auto& w : want) if (strstr(sh.name, w.c_str())) sel = true;
        if (!sel) continue;
        const int M = sh.M, N = sh.N, K = sh.K;
        unsigned short *x, *w, *dy, *y0, *y1, *pre0, *pre1, *dx0, *dx1;
        float *bias, *gw0, *gw1, *ws;
        CK(hipMalloc(&x, (size_t)M * K * 2)); CK(hipMalloc(&w, (size_t)N * K * 2)); CK(hipMalloc(&dy, (size_t)M * N * 2));
        CK(hipMalloc(&y0, (size_t)M * N * 2)); CK(hipMalloc(&y1, (size_t)M * N * 2));
        CK(hipMalloc(&pre0, (size_t)M * N * 2)); CK(hipMalloc(&pre1, (size_t)M * N * 2));
        CK(hipMalloc(&dx0, (size_t)M * K * 2)); CK(hipMalloc(&dx1, (size_t)M * K * 2));
        CK(hipMalloc(&bias, (size_t)N * 4)); CK(hipMalloc(&gw0, (size_t)N * K * 4)); CK(hipMalloc(&gw1, (size_t)N * K * 4));
        fill_bf16(x, (size_t)M * K, g_zero ? 0.f : 1.0f); fill_bf16(w, (size_t)N * K, g_zero ? 0.f : 1.0f / sqrtf((float)K)); fill_bf16(dy, (size_t)M * N, 1.0f);
        { std::vector<float> hb(N); for (auto& v : hb) v = frand(); CK(hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice)); }
        const double fl = 2.0 * M * N * K;
        if (stress > 0) {
            // race screen: the forward (GELU + pre), data-gradient and unsplit weight-gradient kernels are deterministic (no atomics):
            // every repetition must reproduce the first run's outputs bit for bit, whatever the DMA / barrier timing was
            Run rf = {M, N, K, 1, 1, K, K, N, x, w, y1, bias, pre1, 1, nullptr, nullptr, 0, 0, 1, nullptr};
            Run rd = {M, K, N, 1, 0, N, K, K, dy, w, dx1, nullptr, nullptr, 0, nullptr, nullptr, 0, 0, 1, nullptr};
            Run rw = {N, K, M, 0, 0, N, K, K, dy, x, gw1, nullptr, nullptr, 0, nullptr, nullptr, 1, 0, 1, nullptr};
            unsigned long long h0[4] = {0, 0, 0, 0};
            int bad = 0;
            hipStream_t s2; CK(hipStreamCreate(&s2));
            for (int it = 0; it < stress; ++it) {
                if (hog) ecamp_dev_spin(16 + 8 * (it % 13), 64, 20000 + 7000 * (it % 7), (ecampStream_t)s2);   // a different set of CUs held for 10-30 us each time
                CK(hipMemsetAsync(y1, 0xff, (size_t)M * N * 2, s)); CK(hipMemsetAsync(pre1, 0xff, (size_t)M * N * 2, s));
                CK(hipMemsetAsync(dx1, 0xff, (size_t)M * K * 2, s)); CK(hipMemsetAsync(gw1, 0xff, (size_t)N * K * 4, s));
                launch_q8(rf, nslots[0], 0, grid_override, s); launch_q8(rd, nslots[0], 0, grid_override, s); launch_q8(rw, nslots[0], 0, grid_override, s);
                const unsigned long long h[4] = {checksum(y1, (size_t)M * N * 2, s), checksum(pre1, (size_t)M * N * 2, s), checksum(dx1, (size_t)M * K * 2, s),
                                                 checksum(gw1, (size_t)N * K * 4, s)};
                for (int j = 0; j < 4; ++j) { if (it == 0) h0[j] = h[j]; else if (h[j] != h0[j]) { ++bad; printf("  %s: repetition %d output %d differs\n", sh.name, it, j); } }
            }
            CK(hipStreamSynchronize(s2)); CK(hipStreamDestroy(s2));
            printf("%-11s stress %d repetitions of fwd / dgrad / wgrad%s: %s\n", sh.name, stress, hog ? " beside a spinning co-tenant" : "", bad ? "MISMATCH" : "all bit-identical");
            hipFree(x); hipFree(w); hipFree(dy); hipFree(y0); hipFree(y1); hipFree(pre0); hipFree(pre1); hipFree(dx0); hipFree(dx1);
            hipFree(bias); hipFree(gw0); hipFree(gw1);
            continue;
        }
        // ---- forward  Y = gelu(X W^T + b), pre saved
        if (forms & 1) {
            Run r = {M, N, K, 1, 1, K, K, N, x, w, y0, bias, pre0, 1, nullptr, nullptr, 0, 0, 1, nullptr};
            Run q = r; q.C = y1; q.pre = pre1;
            CK(hipMemset(y1, 0xff, (size_t)M * N * 2));
            if (launch_ref(r, 0, s)) { printf("ref failed: %s\n", ecamp_last_error()); return 1; }
            float t_ref = quick ? 0.f : time_us([&] { launch_ref(r, 0, s); });
            launch_ref(r, 0, s);
            printf("%-11s %-5s %6d %6d %5d | %-22s %8.1f %7.0f\n", sh.name, "fwd", M, N, K, "128^2 (r1)", t_ref, fl / t_ref / 1e6);
            for (int ns : nslots)
                for (int dbg : dbgs) {
                    if (plain_only) continue;
                    if (ns != 0 && ns != 4 && ns != 1 && dbg != 0) continue;
                    if (ns == 16 || ns == 12) continue;   // no GELU epilogue in the 16 x 16 x 32 lab kernel
                    CK(hipMemsetAsync(y1, 0xff, (size_t)M * N * 2, s));
                    launch_q8(q, ns, dbg, grid_override, s);
                    CK(hipStreamSynchronize(s));
                    double rn, rn2; double d = compare(y1, y0, (size_t)M * N, false, &rn); double d2 = compare(pre1, pre0, (size_t)M * N, false, &rn2);
                    float t = time_us([&] { launch_q8(q, ns, dbg, grid_override, s); });
                    char v[64]; snprintf(v, sizeof v, "Q8 sch=%d dbg=%d", ns, dbg);
                    printf("%-11s %-5s %6d %6d %5d | %-22s %8.1f %7.0f  maxdiff y %.3g (|y|max %.3g) pre %.3g\n", sh.name, "fwd", M, N, K, v, t, fl / t / 1e6, d, rn, d2);
                }
            // plain (no bias / act / pre): the yardstick form
            Run r2 = {M, N, K, 1, 1, K, K, N, x, w, y0, nullptr, nullptr, 0, nullptr, nullptr, 0, 0, 1, nullptr};
            Run q2 = r2; q2.C = y1;
            launch_ref(r2, 0, s);
            for (int ns : nslots)
            for (int dbg : dbgs) {
                if (ns != 0 && ns != 4 && ns != 1 && ns != 16 && dbg != 0) continue;
                CK(hipMemsetAsync(y1, 0xff, (size_t)M * N * 2, s));
                launch_q8(q2, ns, dbg, grid_override, s);
                CK(hipStreamSynchronize(s));
                double rn; double d = compare(y1, y0, (size_t)M * N, false, &rn);
                float t = time_us([&] { launch_q8(q2, ns, dbg, grid_override, s); });
                char v[64]; snprintf(v, sizeof v, "Q8 plain sch=%d dbg=%d", ns, dbg);
                printf("%-11s %-5s %6d %6d %5d | %-22s %8.1f %7.0f  maxdiff %.3g\n", sh.name, "fwd", M, N, K, v, t, fl / t / 1e6, d);
            }
        }
        // ---- data gradient dX[M,K] = dY[M,N] W[N,K]   (contraction over N; W strided)
        if (forms & 2) {
            Run r = {M, K, N, 1, 0, N, K, K, dy, w, dx0, nullptr, nullptr, 0, nullptr, nullptr, 0, 0, 1, nullptr};
            Run q = r; q.C = dx1;
            if (launch_ref(r, 0, s)) { printf("ref failed: %s\n", ecamp_last_error()); return 1; }
            float t_ref = quick ? 0.f : time_us([&] { launch_ref(r, 0, s); });
            launch_ref(r, 0, s);
            printf("%-11s %-5s %6d %6d %5d | %-22s %8.1f %7.0f\n", sh.name, "dgrad", M, K, N, "128^2 (r1)", t_ref, fl / t_ref / 1e6);
            for (int ns : nslots)
            for (int dbg : dbgs) {   // (dbg 8: the four-wave kernels with conflict-free, WRONG transpose-read addresses -- a timing probe)
                if (dbg != 0 && !(dbg == 8 && (ns == 16 || ns == 12))) continue;
                CK(hipMemsetAsync(dx1, 0xff, (size_t)M * K * 2, s));
                launch_q8(q, ns, dbg, grid_override, s);
                CK(hipStreamSynchronize(s));
                double rn; double d = compare(dx1, dx0, (size_t)M * K, false, &rn);
                float t = time_us([&] { launch_q8(q, ns, dbg, grid_override, s); });
                char v[64]; snprintf(v, sizeof v, "Q8 sch=%d dbg=%d", ns, dbg);
                printf("%-11s %-5s %6d %6d %5d | %-22s %8.1f %7.0f  maxdiff %.3g (max %.3g)\n", sh.name, "dgrad", M, K, N, v, t, fl / t / 1e6, d, rn);
            }
        }
        // ---- weight gradient dW[N,K] (f32) = dY^T[N,M] X[M,K]   (contraction over M; both strided), split-K slabs
        if (forms & 4) {
            int split = ecamp_gemm_suggest_split(N, K, M, 0, 0, ECAMP_BF16);
            CK(hipMalloc(&ws, (size_t)std::max(split, 1) * N * K * 4));
            Run r = {N, K, M, 0, 0, N, K, K, dy, x, gw0, nullptr, nullptr, 0, nullptr, nullptr, 1, 0, split, ws};
            Run q = r; q.C = gw1;
            if (launch_ref(r, 0, s)) { printf("ref failed: %s\n", ecamp_last_error()); return 1; }
            launch_ref(r, 0, s);
            for (int ns : nslots) {
                Run q1 = q; q1.split = 1;     // unsplit: direct f32 store, comparable with the reference
                CK(hipMemsetAsync(gw1, 0xff, (size_t)N * K * 4, s));
                launch_q8(q1, ns, 0, grid_override, s);
                CK(hipStreamSynchronize(s));
                g_ld = K; g_verbose = getenv("LAB_VERBOSE") ? 40 : 0;
                double rn; double d = compare(gw1, gw0, (size_t)N * K, true, &rn);
                g_verbose = 0;
                float t1 = time_us([&] { launch_q8(q1, ns, 0, grid_override, s); });
                float t = time_us([&] { launch_q8(q, ns, 0, grid_override, s); });
                char v[64]; snprintf(v, sizeof v, "Q8 sch=%d", ns);
                printf("%-11s %-5s %6d %6d %5d | %-22s %8.1f %7.0f  split %d slabs only; unsplit %.1f us maxdiff %.3g (max %.3g)\n", sh.name, "wgrad", N, K, M, v, t,
                       fl / t / 1e6, split, t1, d, rn);
            }
            CK(hipFree(ws));
        }
        hipFree(x); hipFree(w); hipFree(dy); hipFree(y0); hipFree(y1); hipFree(pre0); hipFree(pre1); hipFree(dx0); hipFree(dx1);
        hipFree(bias); hipFree(gw0); hipFree(gw1);
    }
    return 0;
}
