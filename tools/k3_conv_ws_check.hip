// k3_conv_ws_check.hip -- GPU tool, NOT part of libdnascent_hip.so: the weight-stationary convolution EXPERIMENT (tools/k3_conv_ws_experiment.h) against k3_conv_split on the device:
// largest difference (the two sum K in different orders: not bit-identical by construction), range words, and the time of both.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
//         -I include -I dnascent_amd/csrc tools/k3_conv_ws_check.hip -o tools/_bin/k3_conv_ws_check
//   tools/_bin/k3_conv_ws_check [rows = 1200128] [taps = 17] [cin = 128] [cout = 256] [add = 1] [iterations = 5]
#define K3_NO_RANGE_CHECK 1
#include "../dnascent_amd/csrc/k3_cnn.hip"
#include "k3_conv_ws_experiment.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

int main(int argc, char **argv) {
    const unsigned R = argc > 1 ? (unsigned)atol(argv[1]) / 256u * 256u : 1200128u;
    const int KW = argc > 2 ? atoi(argv[2]) : 17, CIN = argc > 3 ? atoi(argv[3]) : 128, COUT = argc > 4 ? atoi(argv[4]) : 256, ADD = argc > 5 ? atoi(argv[5]) : 1;
    const int iters = argc > 6 ? atoi(argv[6]) : 5;
    std::mt19937_64 rng(20251003);
    std::normal_distribution<float> N01(0.f, 1.f);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    std::vector<dn_cnn_op> ops(1);
    std::vector<float> blob;
    auto put = [&](size_t n, float sd) { const size_t off = blob.size(); for (size_t i = 0; i < n; i++) blob.push_back(sd * N01(rng)); return (int64_t)off; };
    auto putu = [&](size_t n, float lo, float hi) { const size_t off = blob.size(); for (size_t i = 0; i < n; i++) blob.push_back(lo + (hi - lo) * U(rng)); return (int64_t)off; };
    dn_cnn_op &s = ops[0]; memset(&s, 0, sizeof(s));
    s.op = ADD ? DN_CNN_CONV_ADD : DN_CNN_CONV; s.src = 1; s.dst = 0; s.a = 3; s.k = KW; s.cin = CIN; s.cout = COUT; s.relu = 1;
    s.w = put((size_t)KW * CIN * COUT, sqrtf(2.0f / (KW * CIN))); s.scale = putu(COUT, 0.5f, 0.9f); s.shift = put(COUT, 0.08f);
    // the fp16 pieces as dn_load_cnn lays them out: [step = channel block * k + tap][piece][cout][32], scaled into [2^13, 2^14)
    std::vector<uint16_t> wh; std::vector<int64_t> wh_off(1, 0); std::vector<float> post(1, 1.0f);
    auto f16_bits = [](float f) -> uint16_t { const _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; };
    auto f16_f32 = [](uint16_t b) -> float { _Float16 h; memcpy(&h, &b, 2); return (float)h; };
    {
        const size_t cbn = CIN / 32, steps = (size_t)KW * cbn, blk = (size_t)COUT * 32;
        wh.resize(steps * 2 * blk);
        const float *src = blob.data() + s.w;               // Keras [k][cin][cout]
        float wmax = 0.f;
        for (size_t e = 0; e < (size_t)KW * CIN * COUT; e++) wmax = std::max(wmax, fabsf(src[e]));
        const int up = 13 - ilogbf(wmax);
        const float mul = ldexpf(1.0f, up); post[0] = ldexpf(1.0f, -up);
        for (size_t st = 0; st < steps; st++) {
            const size_t cb = st / (size_t)KW, tp = st % (size_t)KW;
            for (int n = 0; n < COUT; n++)
                for (int kk = 0; kk < 32; kk++) {
                    const float x = src[(tp * CIN + cb * 32 + kk) * COUT + n] * mul;
                    const uint16_t h = f16_bits(x);
                    wh[(st * 2 + 0) * blk + (size_t)n * 32 + kk] = h; wh[(st * 2 + 1) * blk + (size_t)n * 32 + kk] = f16_bits(x - f16_f32(h));
                }
        }
    }
    std::vector<uint8_t> valid(R + 256, 0);
    { unsigned r = 8; while (r + 1000 < R - 8) { unsigned len = 1000 + (unsigned)(U(rng) * 29000); len = std::min(len, R - 8 - r); for (unsigned q = 0; q < len; q++) valid[r + q] = 1; r += len + 8; } }
    std::vector<float> X((size_t)R * CIN, 0.f), Av((size_t)R * COUT, 0.f);
    for (unsigned r = 0; r < R; r++) if (valid[r]) {
        for (int ch = 0; ch < CIN; ch++) X[(size_t)r * CIN + ch] = fmaxf(0.f, N01(rng) * (ch % 7 == 0 ? 3.0f : 1.0f));
        for (int ch = 0; ch < COUT; ch++) Av[(size_t)r * COUT + ch] = N01(rng);
    }
    const size_t bw_ = (size_t)R * std::max(CIN, COUT) * 4;
    float *d_w, *buf[4]; uint16_t *d_wh; uint8_t *d_valid; unsigned *d_range, *d_rowoff, *d_npos; uint64_t *d_iooff; int *d_live;
    CK(hipMalloc((void **)&d_w, blob.size() * 4)); CK(hipMemcpy(d_w, blob.data(), blob.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc((void **)&d_wh, wh.size() * 2)); CK(hipMemcpy(d_wh, wh.data(), wh.size() * 2, hipMemcpyHostToDevice));
    for (int b = 0; b < 4; b++) { CK(hipMalloc((void **)&buf[b], bw_)); CK(hipMemset(buf[b], 0xff, bw_)); }
    CK(hipMemcpy(buf[1], X.data(), X.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(buf[3], Av.data(), Av.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc((void **)&d_valid, R + 256)); CK(hipMemcpy(d_valid, valid.data(), R + 256, hipMemcpyHostToDevice));
    CK(hipMalloc((void **)&d_range, (2 + 2 * 16) * 4)); CK(hipMemset(d_range, 0, (2 + 2 * 16) * 4));
    CK(hipMalloc((void **)&d_rowoff, 64)); CK(hipMalloc((void **)&d_npos, 64)); CK(hipMalloc((void **)&d_iooff, 64)); CK(hipMalloc((void **)&d_live, 256));
    { const unsigned np = R - 16; CK(hipMemcpy(d_npos, &np, 4, hipMemcpyHostToDevice)); const uint64_t z = 0; CK(hipMemcpy(d_iooff, &z, 8, hipMemcpyHostToDevice)); }
    hipStream_t st; CK(hipStreamCreate(&st));
    CnnRun run{};
    run.ops = ops.data(); run.n_ops = 1; run.wts = d_w;
    for (int b = 0; b < 4; b++) run.buf[b] = buf[b];
    run.n_buf = 4;
    run.rows.row_off = d_rowoff; run.rows.valid = d_valid; run.rows.rows = R; run.rows.r0 = 0; run.rows.r1 = 1; run.rows.n_pos = d_npos; run.rows.io_off = d_iooff;
    run.valid = d_valid; run.max_pos = R; run.wts_split = d_wh; run.wb_off = wh_off.data(); run.pieces = 2; run.post = post.data(); run.range_flag = d_range;
    run.n_pass_pos = R - 16; run.row_off_w = d_rowoff; run.live = d_live;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> out[2]; unsigned rw[2][2]; float best[2] = {1e30f, 1e30f};
    for (int m = 0; m < 2; m++) {
        for (int it = 0; it < iters + 1; it++) {
            CK(hipMemsetAsync(buf[0], 0xff, (size_t)R * COUT * 4, st));
            CK(hipMemsetAsync(d_range, 0, (2 + 2 * 16) * 4, st));
            CK(hipEventRecord(e0, st));
            if (m == 0) { if (k3_run(run, st)) { fprintf(stderr, "k3_run failed\n"); return 2; } }
            else {
                CwArgs a{};
                a.X = buf[1]; a.Y = buf[0]; a.Add = ADD ? buf[3] : nullptr; a.valid = d_valid; a.live = d_live; a.rows = (int)R; a.cout = COUT; a.w = d_wh;
                a.scale = d_w + s.scale; a.shift = d_w + s.shift; a.range = d_range + 2; a.post = post[0]; a.relu = s.relu;
                const dim3 g(k3_cu_count()), b(512);
#define CW_GO(KW_, CIN_) do { if (ADD) hipLaunchKernelGGL((k3_conv_ws<KW_, CIN_, true>), g, b, 0, st, a); else hipLaunchKernelGGL((k3_conv_ws<KW_, CIN_, false>), g, b, 0, st, a); } while (0)
                if (KW == 17 && CIN == 128) CW_GO(17, 128); else if (KW == 9 && CIN == 128) CW_GO(9, 128); else if (KW == 9 && CIN == 64) CW_GO(9, 64);
                else if (KW == 3 && CIN == 128) CW_GO(3, 128); else if (KW == 3 && CIN == 64) CW_GO(3, 64); else { fprintf(stderr, "shape not instantiated\n"); return 2; }
            }
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (it) best[m] = std::min(best[m], ms);
        }
        out[m].resize((size_t)R * COUT); CK(hipMemcpy(out[m].data(), buf[0], (size_t)R * COUT * 4, hipMemcpyDeviceToHost));
        unsigned w[4]; CK(hipMemcpy(w, d_range, 16, hipMemcpyDeviceToHost)); rw[m][0] = w[2]; rw[m][1] = w[3];
        char name[128]; if (m == 0) k3_describe(run, 0, name, sizeof(name)); else snprintf(name, sizeof(name), "k3_conv_ws<%d, %d, %s>", KW, CIN, ADD ? "true" : "false");
        const double fl = 2.0 * KW * CIN * COUT * (double)R;
        printf("DN_CNN_CONV_WS=%d  %-36s %8.1f us  (%u rows, %d x %d -> %d%s: %.0f TFLOP/s algorithmic, %.0f issued)\n", m, name, best[m] * 1e3, R, KW, CIN, COUT, ADD ? " + add" : "",
               fl / (best[m] * 1e-3) / 1e12, 3 * fl / (best[m] * 1e-3) / 1e12);
    }
    double maxabs = 0, maxrel = 0, sumsq = 0; size_t nan = 0, big = 0;
    for (size_t i = 0; i < (size_t)R * COUT; i++) {
        const float a = out[0][i], b = out[1][i];
        if (a != a || b != b) { nan++; continue; }
        const double d = fabs((double)a - b);
        maxabs = std::max(maxabs, d); sumsq += (double)a * a;
        if (d > 1e-5 * std::max(1.0, fabs((double)a))) { if (big < 8) printf("  row %zu col %zu: conv_split %.9g  conv_ws %.9g\n", i / COUT, i % COUT, a, b); big++; }
        maxrel = std::max(maxrel, d / std::max(1.0, fabs((double)a)));
    }
    printf("largest |difference| %.3g (relative to max(1, |value|): %.3g; rms of the values %.3g), %zu beyond 1e-5, %zu NaN; range words %08x %08x vs %08x %08x\n", maxabs, maxrel,
           sqrt(sumsq / ((double)R * COUT)), big, nan, rw[0][0], rw[0][1], rw[1][0], rw[1][1]);
#ifdef CW_TRACE
    {
        unsigned long long tr[8][8]; CK(hipMemcpyFromSymbol(tr, HIP_SYMBOL(cw_trace), sizeof(tr)));
        unsigned long long t0 = ~0ull; for (int w = 0; w < 8; w++) if (tr[w][0] && tr[w][0] < t0) t0 = tr[w][0];
        printf("phase stamps of workgroup %d, chunk 40 (ticks after the earliest start): start | first half done (0-3: multiplied, 4-7: reduced + split) | second half done | before barrier | after barrier\n", (int)CW_TRACE);
        for (int w = 0; w < 8; w++) { printf("  wavefront %d", w); for (int i = 0; i < 7; i++) printf(" %6lld", (long long)(tr[w][i] - t0)); printf("\n"); }
    }
#endif
    printf("k3_conv_ws / k3_conv_split: %.2fx\n", best[0] / best[1]);
    printf(big || nan ? "RESULT: MISMATCH\n" : "RESULT: within 1e-5\n");
    return big || nan ? 1 : 0;
}
