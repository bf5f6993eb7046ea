// Development probe: where a workgroup of the head-resident attention BACKWARD kernel spends its life (csrc/attention_bf16.hip built with
// -DATTN_TS: wave 0 stamps the 100 MHz wall clock at the phase boundaries).  On the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DATTN_TS -I ecamp_amd/csrc -o /tmp/attn_phase_probe tools/probes/attn_phase_probe.hip ecamp_amd/csrc/core.hip
//   /tmp/attn_phase_probe
// Stamps: 0 kernel entry | 1 K,V staged (barrier passed) | 2 wave 0 done with phase 1 (dQ) | 3 every wave done with phase 1 |
//         4 Q,dO staged (barrier passed) | 5 wave 0 done with phase 2 (dK, dV)
#include "../../ecamp_amd/csrc/attention_bf16.hip"
#include <algorithm>
#include <vector>

static void run(const char* name, int B, int H, int T, int hd, bool flags) {
    const long D = (long)H * hd, n = (long)B * T * 3 * D;
    bf16_t *qkv, *o, *dO, *dqkv;
    float *lse, *delta;
    int* km;
    unsigned char* bits;
    hipMalloc(&qkv, n * 2); hipMalloc(&dqkv, n * 2); hipMalloc(&o, n / 3 * 2); hipMalloc(&dO, n / 3 * 2);
    hipMalloc(&lse, (long)B * H * T * 4); hipMalloc(&delta, (long)B * H * T * 4); hipMalloc(&km, (long)B * T * 4);
    hipMalloc(&bits, (long)B * H * T * 32);
    std::vector<bf16_t> h(n);
    unsigned s = 12345;
    for (long i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; float f = ((int)(s >> 9) % 2001 - 1000) * 1e-3f; h[i] = (bf16_t)(__builtin_bit_cast(unsigned, f) >> 16); }
    hipMemcpy(qkv, h.data(), n * 2, hipMemcpyHostToDevice);
    hipMemcpy(dO, h.data(), n / 3 * 2, hipMemcpyHostToDevice);
    std::vector<int> hk((long)B * T, 1);
    for (int b = 0; b < B; ++b) for (int t = T / 2 + (b * 7) % (T / 2); t < T; ++t) hk[(long)b * T + t] = 0;
    hipMemcpy(km, hk.data(), (long)B * T * 4, hipMemcpyHostToDevice);
    AttnArgs a;
    memset(&a, 0, sizeof a);
    a.q = qkv; a.k = qkv + D; a.v = qkv + 2 * D; a.o = o; a.dout = dO; a.dq = dqkv; a.dk = dqkv + D; a.dv = dqkv + 2 * D; a.lse = lse; a.delta = delta;
    a.key_mask = flags ? km : nullptr;
    a.q_sb = a.k_sb = a.v_sb = a.dq_sb = a.dk_sb = a.dv_sb = (long)T * 3 * D; a.q_st = a.k_st = a.v_st = a.dq_st = a.dk_st = a.dv_st = 3 * D;
    a.q_sh = a.k_sh = a.v_sh = a.dq_sh = a.dk_sh = a.dv_sh = hd;
    a.o_sb = a.do_sb = (long)T * D; a.o_st = a.do_st = D; a.o_sh = a.do_sh = hd;
    a.B = B; a.H = H; a.Tq = T; a.Tk = T; a.scale = 1.0f / sqrtf((float)hd); a.drop_p = flags ? 0.1f : 0.f; a.seed = 1; a.offset = 2;
    a.drop_bits = flags ? bits : nullptr;
    attn_bf16_fwd(a, hd, 0);
    for (int it = 0; it < 3; ++it) attn_bf16_bwd(a, hd, 0);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    for (int it = 0; it < 10; ++it) attn_bf16_bwd(a, hd, 0);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const int nwg = std::min(B * H, 8192);
    std::vector<long long> ts(8 * 8192);
    hipMemcpyFromSymbol(ts.data(), HIP_SYMBOL(attn_ts_buf), sizeof(long long) * 8 * 8192);
    // per phase: median over workgroups, in microseconds (100 ticks per us)
    const char* ph[5] = {"stage K,V", "phase 1 (wave 0)", "wait for the other waves", "stage Q,dO (+ mask bits)", "phase 2 (wave 0)"};
    printf("%s  B %d H %d T %d hd %d flags %d: %.1f us per launch, %d workgroups\n", name, B, H, T, hd, (int)flags, ms * 100.f, B * H);
    double tot = 0;
    for (int p = 0; p < 5; ++p) {
        std::vector<double> d;
        for (int w = 0; w < nwg; ++w) d.push_back((ts[w * 8 + p + 1] - ts[w * 8 + p]) * 0.01);
        std::sort(d.begin(), d.end());
        printf("    %-28s median %6.2f us   p10 %6.2f   p90 %6.2f\n", ph[p], d[d.size() / 2], d[d.size() / 10], d[d.size() * 9 / 10]);
        tot += d[d.size() / 2];
    }
    std::vector<double> life, start;
    long long t0 = ts[0];
    for (int w = 0; w < nwg; ++w) { life.push_back((ts[w * 8 + 5] - ts[w * 8]) * 0.01); t0 = std::min(t0, ts[w * 8]); }
    std::sort(life.begin(), life.end());
    long long tend = 0;
    for (int w = 0; w < nwg; ++w) tend = std::max(tend, ts[w * 8 + 5]);
    printf("    workgroup lifetime median %.2f us (sum of medians %.2f); first entry to last exit %.1f us\n", life[life.size() / 2], tot, (tend - t0) * 0.01);
    hipFree(qkv); hipFree(dqkv); hipFree(o); hipFree(dO); hipFree(lse); hipFree(delta); hipFree(km); hipFree(bits);
}

int main(int argc, char** argv) {
    const int which = argc > 1 ? atoi(argv[1]) : 0;   // 0 all, 1 report side, 2 decoder, 3 encoder
    if (which == 0 || which == 1) run("report side", 256, 6, 128, 128, true);
    if (which == 0 || which == 2) run("decoder    ", 256, 16, 197, 32, false);
    if (which == 0 || which == 3) run("encoder    ", 256, 12, 50, 64, false);
    return 0;
}
