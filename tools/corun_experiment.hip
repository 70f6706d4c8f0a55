// corun_experiment.hip -- round 4, VERDICT r3 item 1(c): do an MFMA-bound convolution and an HBM-bound separable layer OVERLAY when they
// share the chip, or do they time-slice it?  Two streams, the product kernels themselves (this file includes dnascent_amd/csrc/k3_cnn.hip),
// random operands, 1.2 M rows:
//     A  = k3_conv_split<128, false, 2, 256>  17 taps x 128 -> 256      (the network's largest MFMA-bound layer; 128 VGPRs x 8 wavefronts, 64 KB LDS)
//     B9 = k3_sep_split<128, 9, false, 2>      9 taps, 128 -> 128        (HBM-bound, persistent, 2 workgroups per CU, 234 VGPRs, 64 KB)
//     B5 = k3_sep_split<64, 5, false, 2>       5 taps,  64 ->  64
//     BW = k3_sep_ws<256, 17, false, 2>        17 taps, 256 -> 256       (8 wavefronts x 203 VGPRs + 73 KB: owns a CU)
// For every pair: t(A), t(B) x reps alone, t(A ; B) on one stream, t(A || B) on two streams -- at the product footprints and with both
// capped to HALF a CU (A: extra dynamic LDS so that one workgroup fits; B: a persistent grid of one workgroup per CU).
// speedup = (t(A) + t(B)) / t(A || B): 1.0 = pure time slicing, 2.0 (if t(A) == t(B)) = perfect overlay.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I dnascent_amd/csrc -o tools/_bin/corun tools/corun_experiment.hip
#include "../dnascent_amd/csrc/k3_cnn.hip"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

static unsigned rng_state = 12345u;
static float frand() { rng_state = rng_state * 1664525u + 1013904223u; return ((rng_state >> 8) & 0xffff) / 65536.0f - 0.5f; }
static uint16_t f16bits(float x) { _Float16 h = (_Float16)x; uint16_t u; u = __builtin_bit_cast(uint16_t, h); return u; }

struct Bufs {
    float *X, *Y, *Y2, *Wd, *scale, *shift; uint16_t *Wb; uint8_t *valid; int *live; unsigned *flag; int rows;
};

template <typename F> static float time_ms(F &&f, int reps = 3) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int r = 0; r < reps + 1; r++) {
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, 0);
        f();
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        if (r > 0 && ms < best) best = ms;                 // first repetition = warm-up
    }
    return best;
}

int main(int argc, char **argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 1228800;    // multiple of 256
    int cus = 256; (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    Bufs b; b.rows = rows;
    const size_t act = (size_t)rows * 256;
    (void)hipMalloc(&b.X, act * 4); (void)hipMalloc(&b.Y, act * 4); (void)hipMalloc(&b.Y2, act * 4);
    (void)hipMalloc(&b.Wd, 17 * 256 * 4); (void)hipMalloc(&b.scale, 1024); (void)hipMalloc(&b.shift, 1024);
    const size_t nwb = (size_t)17 * 8 * 2 * 256 * 32;        // enough for 17 taps x 256 channels x 2 pieces x 256 outputs
    (void)hipMalloc(&b.Wb, nwb * 2); (void)hipMalloc(&b.valid, rows + 1024); (void)hipMalloc(&b.live, 4); (void)hipMalloc(&b.flag, 64);
    {
        std::vector<float> h(act);
        for (size_t i = 0; i < act; i++) h[i] = frand() * 2.0f;
        (void)hipMemcpy(b.X, h.data(), act * 4, hipMemcpyHostToDevice);
        std::vector<float> w(17 * 256); for (auto &v : w) v = frand() * 0.5f;
        (void)hipMemcpy(b.Wd, w.data(), w.size() * 4, hipMemcpyHostToDevice);
        std::vector<float> s(256, 1.0f), z(256, 0.01f);
        (void)hipMemcpy(b.scale, s.data(), 1024, hipMemcpyHostToDevice); (void)hipMemcpy(b.shift, z.data(), 1024, hipMemcpyHostToDevice);
        std::vector<uint16_t> wb(nwb); for (auto &v : wb) v = f16bits(frand() * 0.1f);
        (void)hipMemcpy(b.Wb, wb.data(), nwb * 2, hipMemcpyHostToDevice);
        (void)hipMemset(b.valid, 1, rows + 1024);
        (void)hipMemcpy(b.live, &rows, 4, hipMemcpyHostToDevice); (void)hipMemset(b.flag, 0, 64);
    }
    hipStream_t s1, s2; (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    hipEvent_t ev1, ev2; (void)hipEventCreateWithFlags(&ev1, hipEventDisableTiming); (void)hipEventCreateWithFlags(&ev2, hipEventDisableTiming);
    // ---- the launches ----
    auto A = [&](hipStream_t st, unsigned dyn_lds) {         // conv 17 x 128 -> 256, 256-row workgroups
        hipLaunchKernelGGL((k3_conv_split<128, false, 2, 256>), dim3(conv_grid(rows, 256, 128, 256)), dim3(512), dyn_lds, st, b.X, b.Y, b.Wb, b.scale, b.shift,
                           (const float *)nullptr, b.valid, rows, b.live, 17, 128, 256, 1, 1.0f, b.flag);
    };
    auto A3 = [&](hipStream_t st, unsigned dyn_lds) {        // conv 3 x 256 -> 256, 128-row workgroups
        hipLaunchKernelGGL((k3_conv_split<128, false, 2>), dim3(conv_grid(rows, 256, 128)), dim3(256), dyn_lds, st, b.X, b.Y, b.Wb, b.scale, b.shift,
                           (const float *)nullptr, b.valid, rows, b.live, 3, 256, 256, 1, 1.0f, b.flag);
    };
    auto B9 = [&](hipStream_t st, unsigned wgs_per_cu) {
        hipLaunchKernelGGL((k3_sep_split<128, 9, false, 2>), dim3(min(conv_grid(rows, 128, 128), wgs_per_cu * cus)), dim3(256), 0, st, b.X, b.Y2, b.Wd, b.Wb, b.scale,
                           b.shift, (const float *)nullptr, b.valid, rows, b.live, 128, 128, 1, 1.0f, b.flag);
    };
    auto B5 = [&](hipStream_t st, unsigned wgs_per_cu) {
        hipLaunchKernelGGL((k3_sep_split<64, 5, false, 2>), dim3(min(conv_grid(rows, 64, 64), wgs_per_cu * cus)), dim3(256), 0, st, b.X, b.Y2, b.Wd, b.Wb, b.scale,
                           b.shift, (const float *)nullptr, b.valid, rows, b.live, 64, 64, 1, 1.0f, b.flag);
    };
    auto BW = [&](hipStream_t st, unsigned) {
        hipLaunchKernelGGL((k3_sep_ws<256, 17, false, 2>), dim3(min(conv_grid(rows, 256, 256), (unsigned)cus)), dim3(512), 0, st, b.X, b.Y2, b.Wd, b.Wb, b.scale,
                           b.shift, (const float *)nullptr, b.valid, rows, b.live, 256, 256, 1, 1.0f, b.flag);
    };
    struct Case { const char *name; int which_a, which_b; unsigned a_lds; unsigned b_wgs; int b_reps; };
    const Case cases[] = {
        {"conv17x128->256 (2 WG/CU) || sep9x128 x3 (2 WG/CU): product footprints", 0, 0, 0, 2, 3},
        {"conv17x128->256 (1 WG/CU) || sep9x128 x3 (1 WG/CU): half a CU each", 0, 0, 24 * 1024, 1, 3},
        {"conv17x128->256 (2 WG/CU) || sep9x128 x3 (1 WG/CU)", 0, 0, 0, 1, 3},
        {"conv17x128->256 (1 WG/CU) || sep9x128 x3 (2 WG/CU)", 0, 0, 24 * 1024, 2, 3},
        {"conv17x128->256 (2 WG/CU) || sep5x64 x6 (3 WG/CU)", 0, 1, 0, 3, 6},
        {"conv17x128->256 (1 WG/CU) || sep5x64 x6 (1 WG/CU)", 0, 1, 24 * 1024, 1, 6},
        {"conv17x128->256 (1 WG/CU) || sep5x64 x6 (2 WG/CU)", 0, 1, 24 * 1024, 2, 6},
        {"conv17x128->256 (2 WG/CU) || sep_ws17x256 x4", 0, 2, 0, 1, 4},
        {"conv17x128->256 (1 WG/CU) || sep_ws17x256 x4", 0, 2, 24 * 1024, 1, 4},
        {"conv3x256->256 (2 WG/CU of 128 rows) || sep9x128 x2 (2 WG/CU)", 1, 0, 0, 2, 2},
        {"conv3x256->256 (1 WG/CU) || sep9x128 x2 (1 WG/CU)", 1, 0, 48 * 1024, 1, 2},
    };
    printf("rows %d, CUs %d\n", rows, cus);
    if (argc > 2 && !strcmp(argv[2], "mall")) {
        // Does a layer run faster when its input and output stay in the 256 MiB Infinity Cache?  The same separable layer ping-pongs between two
        // buffers over the first `chunk` rows only, 12 launches back to back; per-launch time and layer I/O rate against the chunk size.
        const int chunks[] = {16384, 32768, 65536, 131072, 262144, 524288, 1228800};
        for (int which = 0; which < 3; which++) {
            for (int chunk : chunks) {
                if (chunk > rows) continue;
                (void)hipMemcpy(b.live, &chunk, 4, hipMemcpyHostToDevice);
                const int cin = which == 0 ? 128 : which == 1 ? 64 : 256, reps = 12;
                auto go = [&](const float *in, float *out) {
                    if (which == 0) hipLaunchKernelGGL((k3_sep_split<128, 9, false, 2>), dim3(min(conv_grid(chunk, 128, 128), 2u * cus)), dim3(256), 0, s1, in, out, b.Wd, b.Wb, b.scale,
                                                       b.shift, (const float *)nullptr, b.valid, chunk, b.live, 128, 128, 1, 1.0f, b.flag);
                    else if (which == 1) hipLaunchKernelGGL((k3_sep_split<64, 5, false, 2>), dim3(min(conv_grid(chunk, 64, 64), 3u * cus)), dim3(256), 0, s1, in, out, b.Wd, b.Wb, b.scale,
                                                       b.shift, (const float *)nullptr, b.valid, chunk, b.live, 64, 64, 1, 1.0f, b.flag);
                    else hipLaunchKernelGGL((k3_sep_ws<256, 17, false, 2>), dim3(min(conv_grid(chunk, 256, 256), (unsigned)cus)), dim3(512), 0, s1, in, out, b.Wd, b.Wb, b.scale,
                                                       b.shift, (const float *)nullptr, b.valid, chunk, b.live, 256, 256, 1, 1.0f, b.flag);
                };
                const float t = time_ms([&] {
                    (void)hipEventRecord(ev1, 0); (void)hipStreamWaitEvent(s1, ev1, 0);
                    for (int r = 0; r < reps; r++) { if (r & 1) go(b.Y2, b.X); else go(b.X, b.Y2); }
                    (void)hipEventRecord(ev1, s1); (void)hipStreamWaitEvent(0, ev1, 0);
                });
                const double bytes = (double)chunk * cin * 4.0 * 2.0;
                printf("%-18s chunk %8d rows (%6.1f MB in + out)  %8.2f us per launch  %7.0f GB/s of layer I/O  %6.2f ns per row\n",
                       which == 0 ? "sep9 128->128" : which == 1 ? "sep5 64->64" : "sep_ws17 256->256", chunk, bytes / 1e6, t * 1e3 / reps, bytes / (t * 1e-3 / reps) / 1e9, t * 1e6 / reps / chunk);
            }
        }
        return 0;
    }
    for (const Case &c : cases) {
        auto la = [&](hipStream_t st) { if (c.which_a == 0) A(st, c.a_lds); else A3(st, c.a_lds); };
        auto lb = [&](hipStream_t st) { for (int r = 0; r < c.b_reps; r++) { if (c.which_b == 0) B9(st, c.b_wgs); else if (c.which_b == 1) B5(st, c.b_wgs); else BW(st, c.b_wgs); } };
        // every timing brackets on the default stream: fork to s1 / s2 with events, join back
        auto on = [&](auto &&body) {
            return time_ms([&] {
                (void)hipEventRecord(ev1, 0); (void)hipStreamWaitEvent(s1, ev1, 0); (void)hipStreamWaitEvent(s2, ev1, 0);
                body();
                (void)hipEventRecord(ev1, s1); (void)hipEventRecord(ev2, s2); (void)hipStreamWaitEvent(0, ev1, 0); (void)hipStreamWaitEvent(0, ev2, 0);
            });
        };
        const float ta = on([&] { la(s1); });
        const float tb = on([&] { lb(s1); });
        const float tseq = on([&] { la(s1); lb(s1); });
        const float tpar = on([&] { la(s1); lb(s2); });
        const float tpar2 = on([&] { lb(s2); la(s1); });
        printf("%-78s tA %7.3f  tB %7.3f  seq %7.3f  A||B %7.3f  B||A %7.3f ms  speedup %.3f\n", c.name, ta, tb, tseq, tpar, tpar2, (ta + tb) / fminf(tpar, tpar2));
    }
    return 0;
}
