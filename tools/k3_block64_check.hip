// k3_block64_check.hip -- GPU tool, NOT part of libdnascent_hip.so: the fused 64-channel residual block (csrc/k3_block64.h) against the
// layer-by-layer kernels it replaces, on the device, bit for bit, and the time of both.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
//         -I include -I dnascent_amd/csrc tools/k3_block64_check.hip -o tools/_bin/k3_block64_check
//   tools/_bin/k3_block64_check [rows = 1200128] [iterations = 5] [grid override]
// One block of the default description's shape (cnn_model.block(cur = 1, k = 5, 64 -> 64)): ops 0-11 = six (DWCONV, CONV 1 x 1) pairs ping-ponging
// between buffers 2 and 3, op 12 = CONV_ADD 5 x 64 -> 64 reading buffer 1, adding buffer 3, writing buffer 0.  Random weights and activations, a
// validity mask with padding rows between "reads" of random length.  Runs k3_run three times per setting -- DN_CNN_BLOCK64 = 0 (layer by layer),
// 2 (whole block in one launch), 1 (the six separable layers in one launch, the shortcut on its own) -- and compares outputs and range reports.
#define K3_NO_RANGE_CHECK 1                                 /* keep the per-op range words for the comparison */
#include "../dnascent_amd/csrc/k3_cnn.hip"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

int main(int argc, char **argv) {
    const unsigned R = argc > 1 ? (unsigned)atol(argv[1]) / 256u * 256u : 1200128u;
    const int iters = argc > 2 ? atoi(argv[2]) : 5;
    std::mt19937_64 rng(20251003);
    std::normal_distribution<float> N01(0.f, 1.f);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    // ---- description: 13 ops + fp32 blob (Keras layouts), as cnn_model.block builds them ----
    std::vector<dn_cnn_op> ops(13);
    std::vector<float> blob;
    auto put = [&](size_t n, float sd, float mean = 0.f) { const size_t off = blob.size(); for (size_t i = 0; i < n; i++) blob.push_back(mean + sd * N01(rng)); return (int64_t)off; };
    auto putu = [&](size_t n, float lo, float hi) { const size_t off = blob.size(); for (size_t i = 0; i < n; i++) blob.push_back(lo + (hi - lo) * U(rng)); return (int64_t)off; };
    for (int j = 0; j < 6; j++) {
        dn_cnn_op &d = ops[2 * j], &p = ops[2 * j + 1];
        memset(&d, 0, sizeof(d)); memset(&p, 0, sizeof(p));
        d.op = DN_CNN_DWCONV; d.src = j == 0 ? 1 : 3; d.dst = 2; d.k = 5; d.cin = 64; d.cout = 64; d.w = put(5 * 64, sqrtf(1.0f / 5));
        p.op = DN_CNN_CONV; p.src = 2; p.dst = 3; p.k = 1; p.cin = 64; p.cout = 64; p.relu = j < 5; p.a = 0; p.b = 0;
        p.w = put(64 * 64, sqrtf(2.0f / 64)); p.scale = putu(64, j < 5 ? 0.7f : 0.5f, j < 5 ? 1.3f : 0.9f); p.shift = put(64, 0.08f);
    }
    { dn_cnn_op &s = ops[12]; memset(&s, 0, sizeof(s));
      s.op = DN_CNN_CONV_ADD; s.src = 1; s.dst = 0; s.a = 3; s.k = 5; s.cin = 64; s.cout = 64; s.relu = 1;
      s.w = put(5 * 64 * 64, sqrtf(2.0f / 320)); s.scale = putu(64, 0.5f, 0.9f); s.shift = put(64, 0.08f); }
    // ---- the fp16 pieces of every convolution, as dn_load_cnn lays them out: [step = channel block * k + tap][piece][cout][32], scaled into [2^13, 2^14) ----
    std::vector<uint16_t> wh; std::vector<int64_t> wh_off(13, 0); std::vector<float> post(13, 1.0f);
    auto f16_bits = [](float f) -> uint16_t { const _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; };
    auto f16_f32 = [](uint16_t b) -> float { _Float16 h; memcpy(&h, &b, 2); return (float)h; };
    for (int i = 0; i < 13; i++) {
        const dn_cnn_op &o = ops[i];
        if (o.op != DN_CNN_CONV && o.op != DN_CNN_CONV_ADD) continue;
        wh_off[i] = (int64_t)wh.size();
        const size_t cbn = o.cin / 32, steps = (size_t)o.k * cbn, blk = (size_t)o.cout * 32;
        wh.resize(wh.size() + steps * 2 * blk);
        uint16_t *dst = wh.data() + wh_off[i];
        const float *src = blob.data() + o.w;               // Keras [k][cin][cout]
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
    // ---- rows: 8 leading padding rows, then "reads" of 1 000 .. 30 000 rows with 8 padding rows between them; X = N(0, 1) on live rows, 0 elsewhere ----
    std::vector<uint8_t> valid(R + 256, 0);
    { unsigned r = 8; while (r + 1000 < R - 8) { unsigned len = 1000 + (unsigned)(U(rng) * 29000); len = std::min(len, R - 8 - r); for (unsigned q = 0; q < len; q++) valid[r + q] = 1; r += len + 8; } }
    std::vector<float> X((size_t)R * 64, 0.f);
    for (unsigned r = 0; r < R; r++) if (valid[r]) for (int ch = 0; ch < 64; ch++) X[(size_t)r * 64 + ch] = N01(rng) * (ch % 7 == 0 ? 3.0f : 1.0f);
    // ---- device ----
    float *d_w, *buf[4]; uint16_t *d_wh; uint8_t *d_valid; unsigned *d_range, *d_rowoff, *d_npos; uint64_t *d_iooff; int *d_live;
    CK(hipMalloc((void **)&d_w, blob.size() * 4)); CK(hipMemcpy(d_w, blob.data(), blob.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc((void **)&d_wh, wh.size() * 2)); CK(hipMemcpy(d_wh, wh.data(), wh.size() * 2, hipMemcpyHostToDevice));
    for (int b = 0; b < 4; b++) { CK(hipMalloc((void **)&buf[b], (size_t)R * 64 * 4)); CK(hipMemset(buf[b], 0xff, (size_t)R * 64 * 4)); }     // NaN-filled: a row nobody wrote shows
    CK(hipMalloc((void **)&d_valid, R + 256)); CK(hipMemcpy(d_valid, valid.data(), R + 256, hipMemcpyHostToDevice));
    CK(hipMalloc((void **)&d_range, (2 + 2 * 16) * 4)); CK(hipMemset(d_range, 0, (2 + 2 * 16) * 4));
    CK(hipMalloc((void **)&d_rowoff, 64)); CK(hipMalloc((void **)&d_npos, 64)); CK(hipMalloc((void **)&d_iooff, 64)); CK(hipMalloc((void **)&d_live, 256));
    { const unsigned np = R - 16; CK(hipMemcpy(d_npos, &np, 4, hipMemcpyHostToDevice)); const uint64_t z = 0; CK(hipMemcpy(d_iooff, &z, 8, hipMemcpyHostToDevice)); }
    hipStream_t st; CK(hipStreamCreate(&st));
    CnnRun run{};
    run.ops = ops.data(); run.n_ops = 13; run.wts = d_w;
    for (int b = 0; b < 4; b++) run.buf[b] = buf[b];
    run.n_buf = 4;
    run.rows.row_off = d_rowoff; run.rows.valid = d_valid; run.rows.rows = R; run.rows.r0 = 0; run.rows.r1 = 1; run.rows.n_pos = d_npos; run.rows.io_off = d_iooff;
    run.valid = d_valid; run.max_pos = R; run.wts_split = d_wh; run.wb_off = wh_off.data(); run.pieces = 2; run.post = post.data(); run.range_flag = d_range;
    run.n_pass_pos = R - 16; run.row_off_w = d_rowoff; run.live = d_live;
    if (argc > 3) setenv("DN_CNN_WS_WGS", argv[3], 1);       // grid override of the persistent kernels (k3_cu_count)
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> out[3]; std::vector<unsigned> rng_words[3]; float best[3] = {1e30f, 1e30f, 1e30f};
    const int modes[3] = {0, 2, 1};
    for (int m = 0; m < 3; m++) {
        k3_block64_force = modes[m];
        for (int it = 0; it < iters + 1; it++) {
            CK(hipMemcpyAsync(buf[1], X.data(), (size_t)R * 64 * 4, hipMemcpyHostToDevice, st));
            for (int b : {0, 2, 3}) CK(hipMemsetAsync(buf[b], 0xff, (size_t)R * 64 * 4, st));
            CK(hipMemsetAsync(d_range, 0, (2 + 2 * 16) * 4, st));
            CK(hipEventRecord(e0, st));
            if (k3_run(run, st)) { fprintf(stderr, "k3_run failed\n"); return 2; }
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (it) best[m] = std::min(best[m], ms);
        }
        out[m].resize((size_t)R * 64); CK(hipMemcpy(out[m].data(), buf[0], (size_t)R * 64 * 4, hipMemcpyDeviceToHost));
        rng_words[m].resize(2 + 2 * 16); CK(hipMemcpy(rng_words[m].data(), d_range, (2 + 2 * 16) * 4, hipMemcpyDeviceToHost));
        char name[128]; k3_describe(run, 0, name, sizeof(name));
        printf("DN_CNN_BLOCK64=%d  op 0 takes %-22s  13 ops in %8.1f us  (%u rows, %.0f GB/s of block I/O at 512 B per row)\n", modes[m], name, best[m] * 1e3, R,
               (double)R * 512 / (best[m] * 1e-3) / 1e9);
    }
    int bad = 0;
    for (int m = 1; m < 3; m++) {
        size_t diff = 0, nan = 0; int shown = 0;
        for (size_t i = 0; i < (size_t)R * 64; i++) {
            unsigned a, b; memcpy(&a, &out[0][i], 4); memcpy(&b, &out[m][i], 4);
            if (a != b) {
                diff++;
                if (shown < 12) { printf("  mode %d differs at row %zu col %zu (valid %d): layer-by-layer %.9g  fused %.9g\n", modes[m], i / 64, i % 64, valid[i / 64], out[0][i], out[m][i]); shown++; }
            }
            if (out[m][i] != out[m][i]) nan++;
        }
        bool rsame = true;
        for (int w = 2; w < 2 + 2 * 13; w++) if (rng_words[0][w] != rng_words[m][w]) { rsame = false; printf("  mode %d range word %d (op %d, %s): %08x vs %08x\n", modes[m], w, (w - 2) / 2, (w & 1) ? "largest |value|" : "overflow flag", rng_words[0][w], rng_words[m][w]); }
        printf("mode %d vs layer by layer: %zu of %zu values differ, %zu NaN, range report %s\n", modes[m], diff, (size_t)R * 64, nan, rsame ? "identical" : "DIFFERENT");
        bad += diff != 0 || !rsame;
    }
#ifdef B64_TRACE
    {   // phase stamps of workgroup B64_TRACE at step B64_TRACE_STEP of the last whole-block run (mode 2 ran second: rerun it for the stamps)
        k3_block64_force = 2;
        CK(hipMemcpyAsync(buf[1], X.data(), (size_t)R * 64 * 4, hipMemcpyHostToDevice, st));
        if (k3_run(run, st)) return 2;
        CK(hipStreamSynchronize(st));
        unsigned long long tr[8][16];
        CK(hipMemcpyFromSymbol(tr, HIP_SYMBOL(b64_trace), sizeof(tr)));
        const char *names[8] = {"stage 0", "stage 1", "stage 2", "stage 3", "stage 4", "stage 5", "conv 0", "conv 1"};
        unsigned long long t0 = ~0ull;
        for (int r = 0; r < 8; r++) if (tr[r][0] && tr[r][0] < t0) t0 = tr[r][0];
        printf("phase stamps (ticks after the earliest step start): begin | cb0: inputs ready, filter done, planes stored, MFMAs issued | cb1: ... | before barrier, after barrier\n");
        for (int r = 0; r < 8; r++) {
            printf("  %-8s", names[r]);
            for (int i = 0; i < 11; i++) printf(" %7lld", tr[r][i] ? (long long)(tr[r][i] - t0) : -1ll);
            printf("\n");
        }
        printf("  workgroup %d: %llu steps in %llu ticks = %.0f ticks per step (first barrier to last)\n", (int)B64_TRACE, tr[0][14] - 1, tr[0][13] - tr[0][12], (double)(tr[0][13] - tr[0][12]) / (double)(tr[0][14] - 1));
    }
#endif
    printf("speed-up of the block: %.2fx (whole block), %.2fx (separable layers fused, shortcut apart)\n", best[0] / best[1], best[0] / best[2]);
    printf(bad ? "RESULT: MISMATCH\n" : "RESULT: bit-identical\n");
    return bad ? 1 : 0;
}
