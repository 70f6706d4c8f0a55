// k3_one256_check.hip -- GPU tool, NOT part of libdnascent_hip.so: the un-split 17-tap separable layer of the 256-channel stage (tools/k3_one256_experiment.h: an experiment that lost) against the layer-by-layer kernels
// they replace, on the device, bit for bit, and the time of both.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
//         -I include -I dnascent_amd/csrc tools/k3_one256_check.hip -o tools/_bin/k3_one256_check
//   tools/_bin/k3_one256_check [rows = 1200128] [cin0 = 256 | 128] [iterations = 5] [layers = 1]
#define K3_NO_RANGE_CHECK 1
#include "../dnascent_amd/csrc/k3_cnn.hip"
#include "k3_one256_experiment.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

int main(int argc, char **argv) {
    const unsigned R = argc > 1 ? (unsigned)atol(argv[1]) / 256u * 256u : 1200128u;
    const int CIN0 = argc > 2 ? atoi(argv[2]) : 256;
    const int iters = argc > 3 ? atoi(argv[3]) : 5;
    const int NL = argc > 4 ? atoi(argv[4]) : 1;             // separable layers
    std::mt19937_64 rng(20251003);
    std::normal_distribution<float> N01(0.f, 1.f);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    const int NOPS = 2 * NL;
    std::vector<dn_cnn_op> ops(NOPS);
    std::vector<float> blob;
    auto put = [&](size_t n, float sd, float mean = 0.f) { const size_t off = blob.size(); for (size_t i = 0; i < n; i++) blob.push_back(mean + sd * N01(rng)); return (int64_t)off; };
    auto putu = [&](size_t n, float lo, float hi) { const size_t off = blob.size(); for (size_t i = 0; i < n; i++) blob.push_back(lo + (hi - lo) * U(rng)); return (int64_t)off; };
    for (int j = 0; j < NL; j++) {                          // as cnn_model.block builds a chain: depthwise src -> 2, pointwise 2 -> 3; the chain's input is buffer 1
        dn_cnn_op &d = ops[2 * j], &p = ops[2 * j + 1];
        memset(&d, 0, sizeof(d)); memset(&p, 0, sizeof(p));
        const int cin = j == 0 ? CIN0 : 256;
        d.op = DN_CNN_DWCONV; d.src = j == 0 ? 1 : 3; d.dst = 2; d.k = 17; d.cin = cin; d.cout = cin; d.w = put(17 * cin, sqrtf(1.0f / 17));
        p.op = DN_CNN_CONV; p.src = 2; p.dst = 3; p.k = 1; p.cin = cin; p.cout = 256; p.relu = j + 1 < NL;
        p.w = put((size_t)cin * 256, sqrtf(2.0f / cin)); p.scale = putu(256, 0.7f, 1.3f); p.shift = put(256, 0.08f);
    }
    std::vector<uint16_t> wh; std::vector<int64_t> wh_off(NOPS, 0); std::vector<float> post(NOPS, 1.0f);
    auto f16_bits = [](float f) -> uint16_t { const _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; };
    auto f16_f32 = [](uint16_t b) -> float { _Float16 h; memcpy(&h, &b, 2); return (float)h; };
    for (int i = 0; i < NOPS; i++) {
        const dn_cnn_op &o = ops[i];
        if (o.op != DN_CNN_CONV) continue;
        wh_off[i] = (int64_t)wh.size();
        const size_t cbn = o.cin / 32, steps = (size_t)o.k * cbn, blk = (size_t)o.cout * 32;
        wh.resize(wh.size() + steps * 2 * blk);
        uint16_t *dst = wh.data() + wh_off[i];
        const float *src = blob.data() + o.w;
        float wmax = 0.f;
        for (size_t e = 0; e < (size_t)o.k * o.cin * o.cout; e++) wmax = std::max(wmax, fabsf(src[e]));
        const int up = 13 - ilogbf(wmax);
        const float mul = ldexpf(1.0f, up); post[i] = ldexpf(1.0f, -up);
        for (size_t st = 0; st < steps; st++) {
            const size_t cb = st / (size_t)o.k, tp = st % (size_t)o.k;
            for (int n = 0; n < o.cout; n++)
                for (int kk = 0; kk < 32; kk++) {
                    const float x = src[(tp * o.cin + cb * 32 + kk) * o.cout + n] * mul;
                    const uint16_t h = f16_bits(x);
                    dst[(st * 2 + 0) * blk + (size_t)n * 32 + kk] = h; dst[(st * 2 + 1) * blk + (size_t)n * 32 + kk] = f16_bits(x - f16_f32(h));
                }
        }
    }
    std::vector<uint8_t> valid(R + 256, 0);
    { unsigned r = 8; while (r + 1000 < R - 8) { unsigned len = 1000 + (unsigned)(U(rng) * 29000); len = std::min(len, R - 8 - r); for (unsigned q = 0; q < len; q++) valid[r + q] = 1; r += len + 8; } }
    std::vector<float> X((size_t)R * CIN0, 0.f);
    for (unsigned r = 0; r < R; r++) if (valid[r]) for (int ch = 0; ch < CIN0; ch++) X[(size_t)r * CIN0 + ch] = fmaxf(0.f, N01(rng) * (ch % 7 == 0 ? 3.0f : 1.0f));
    float *d_w, *buf[4]; uint16_t *d_wh; uint8_t *d_valid; unsigned *d_range, *d_rowoff, *d_npos; uint64_t *d_iooff; int *d_live;
    const size_t bb = (size_t)R * 256 * 4;
    CK(hipMalloc((void **)&d_w, blob.size() * 4)); CK(hipMemcpy(d_w, blob.data(), blob.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc((void **)&d_wh, wh.size() * 2)); CK(hipMemcpy(d_wh, wh.data(), wh.size() * 2, hipMemcpyHostToDevice));
    for (int b = 0; b < 4; b++) { CK(hipMalloc((void **)&buf[b], bb)); CK(hipMemset(buf[b], 0xff, bb)); }
    CK(hipMalloc((void **)&d_valid, R + 256)); CK(hipMemcpy(d_valid, valid.data(), R + 256, hipMemcpyHostToDevice));
    CK(hipMalloc((void **)&d_range, (2 + 2 * 16) * 4)); CK(hipMemset(d_range, 0, (2 + 2 * 16) * 4));
    CK(hipMalloc((void **)&d_rowoff, 64)); CK(hipMalloc((void **)&d_npos, 64)); CK(hipMalloc((void **)&d_iooff, 64)); CK(hipMalloc((void **)&d_live, 256));
    { const unsigned np = R - 16; CK(hipMemcpy(d_npos, &np, 4, hipMemcpyHostToDevice)); const uint64_t z = 0; CK(hipMemcpy(d_iooff, &z, 8, hipMemcpyHostToDevice)); }
    hipStream_t st; CK(hipStreamCreate(&st));
    CnnRun run{};
    run.ops = ops.data(); run.n_ops = NOPS; run.wts = d_w;
    for (int b = 0; b < 4; b++) run.buf[b] = buf[b];
    run.n_buf = 4;
    run.rows.row_off = d_rowoff; run.rows.valid = d_valid; run.rows.rows = R; run.rows.r0 = 0; run.rows.r1 = 1; run.rows.n_pos = d_npos; run.rows.io_off = d_iooff;
    run.valid = d_valid; run.max_pos = R; run.wts_split = d_wh; run.wb_off = wh_off.data(); run.pieces = 2; run.post = post.data(); run.range_flag = d_range;
    run.n_pass_pos = R - 16; run.row_off_w = d_rowoff; run.live = d_live;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> out[2]; std::vector<unsigned> rng_words[2]; float best[2] = {1e30f, 1e30f};
    // where the result lands: the walker swaps buffer pointers per fused launch; ask it by running on poisoned buffers and looking which one is finite on a live row
    for (int m = 0; m < 2; m++) {
        for (int it = 0; it < iters + 1; it++) {
            CK(hipMemcpyAsync(buf[1], X.data(), X.size() * 4, hipMemcpyHostToDevice, st));
            for (int b : {0, 2, 3}) CK(hipMemsetAsync(buf[b], 0xff, bb, st));
            CK(hipMemsetAsync(d_range, 0, (2 + 2 * 16) * 4, st));
            CK(hipEventRecord(e0, st));
            if (m == 0) { if (k3_run(run, st)) { fprintf(stderr, "k3_run failed\n"); return 2; } }
            else {
                O256Args a{};
                a.X = buf[1]; a.Y = buf[2]; a.valid = d_valid; a.live = d_live; a.rows = (int)R;
                a.wd = d_w + ops[0].w; a.wb = d_wh + wh_off[1]; a.scale = d_w + ops[1].scale; a.shift = d_w + ops[1].shift; a.range = d_range + 2; a.post = post[1]; a.relu = ops[1].relu;
                const unsigned grid = std::max(1u, std::min(k3_cu_count(), R / 32u));
                if (CIN0 == 128) hipLaunchKernelGGL((k3_one256<128>), dim3(grid), dim3(256), 0, st, a); else hipLaunchKernelGGL((k3_one256<256>), dim3(grid), dim3(256), 0, st, a);
            }
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (it) best[m] = std::min(best[m], ms);
        }
        // layer by layer: NL swaps of (2, 3): even -> the result is in buffer 3's memory; fused: NL / 2 swaps
        const int swaps = NL;                                   // one pointer swap per layer on either path
        const int where = (swaps & 1) ? 2 : 3;
        out[m].resize((size_t)R * 256); CK(hipMemcpy(out[m].data(), buf[where], bb, hipMemcpyDeviceToHost));
        rng_words[m].resize(2 + 2 * 16); CK(hipMemcpy(rng_words[m].data(), d_range, (2 + 2 * 16) * 4, hipMemcpyDeviceToHost));
        char name[128]; if (m == 0) k3_describe(run, 0, name, sizeof(name)); else snprintf(name, sizeof(name), "k3_one256<%d>", CIN0);
        printf("DN_CNN_ONE256=%d  op 0 takes %-30s %d layers in %8.1f us  (%u rows, %.0f GB/s of layer I/O at %d B per row and layer)\n", m, name, NL, best[m] * 1e3, R,
               (double)R * (CIN0 * 4 + 256 * 4 + (NL - 1) * 2048.0) / (best[m] * 1e-3) / 1e9, 2048);
    }
    size_t diff = 0, nan = 0; int shown = 0;
    for (size_t i = 0; i < (size_t)R * 256; i++) {
        unsigned a, b; memcpy(&a, &out[0][i], 4); memcpy(&b, &out[1][i], 4);
        if (a != b) { diff++; if (shown < 12) { printf("  differs at row %zu col %zu (valid %d): layer-by-layer %.9g  fused %.9g\n", i / 256, i % 256, valid[i / 256], out[0][i], out[1][i]); shown++; } }
        if (out[1][i] != out[1][i]) nan++;
    }
    bool rsame = true;
    for (int w = 2; w < 2 + 2 * NOPS && w < 2 + 2 * 16; w++) if (rng_words[0][w] != rng_words[1][w]) { rsame = false; printf("  range word %d (op %d): %08x vs %08x\n", w, (w - 2) / 2, rng_words[0][w], rng_words[1][w]); }
    printf("fused vs layer by layer: %zu of %zu values differ, %zu NaN, range report %s\n", diff, (size_t)R * 256, nan, rsame ? "identical" : "DIFFERENT");
#ifdef O256_TRACE
    {
        unsigned long long tr[4][8]; CK(hipMemcpyFromSymbol(tr, HIP_SYMBOL(o256_trace), sizeof(tr)));
        unsigned long long t0 = ~0ull; for (int w = 0; w < 4; w++) if (tr[w][0] && tr[w][0] < t0) t0 = tr[w][0];
        printf("phase stamps of workgroup %d, step 40 (ticks after the earliest start): 0 start | 1 multiplied | 2 exchanged (wavefronts 0, 1) / multiplied (2, 3) | 3 filtered | 4 planes stored / rows stored | 5 after the barrier | 6 epilogue done | 7 main exchange done\n", (int)O256_TRACE);
        for (int w = 0; w < 4; w++) { printf("  wavefront %d", w); for (int i = 0; i < 5; i++) printf(" %6lld", tr[w][i] ? (long long)(tr[w][i] - t0) : -1ll); printf("\n"); }
    }
#endif
    printf("speed-up: %.2fx\n", best[0] / best[1]);
    printf(diff || !rsame ? "RESULT: MISMATCH\n" : "RESULT: bit-identical\n");
    return diff || !rsame ? 1 : 0;
}
