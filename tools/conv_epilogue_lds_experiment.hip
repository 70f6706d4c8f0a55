// conv_epilogue_lds_experiment.hip -- NOT part of libdnascent_hip.so.  Round-4 experiment, kept as the record of a negative result
// (moved out of k3_cnn.hip in round 5: round-4 verdict item 7).  The fragment needs k3_cnn.hip around it (f32x16, f32x4, conv_epilogue's
// callers); it was switched in with -DCONV_EP_LDS=1 (k3_conv_split: the dead A planes as scratch) / -DSEP_EP_LDS=1 (k3_sep_split<128, ., ., 2>: 16 KB
// of extra LDS).  Result (gpurun_out/r4u): bit-identical, every layer within 1 % of its time, the pipeline 762-771 against 770-775 Msamples/s:
// the interval the phase trace showed between the last step and the end of the epilogue is the drain of the stores' memory latency, not their issue.
//
// call sites as they were:
//   k3_conv_split:  float *scr = reinterpret_cast<float *>(&As[0][0]) + (tid_e >> 6) * (32 * 40);
//                   conv_epilogue_lds<BN, ADD, 40>(acc, Y, scale, shift, Add, valid, m0, n0, tid_e >> 7, (tid_e >> 6) & 1, tid_e & 63, cout, relu, post, scr);
//   k3_sep_split:   __shared__ float Ep[4 * 32 * 32];
//                   conv_epilogue_lds<BN, ADD, 32>(acc, Y, scale, shift, Add, valid, tile_m0(it), n0, wm, wn, lane, cout, relu, post, Ep + wave * (32 * 32));

// (Round 4) the same epilogue with its stores THROUGH LDS: every 32 x 32 accumulator tile is written to a per-wavefront scratch in accumulator layout (lane =
// column) and read back as 16-byte row pieces, so that a tile leaves as 4 x dwordx4 stores -- 8 rows x 128 bytes each -- instead of 16 x dword (2 rows x 128 bytes
// each).  The CU's address unit takes a 64-lane store every ~16 clocks whatever its width: a 128 x 128 tile's 4 x 64 dword stores were a third of a k3_sep_split
// tile under the tracer (profiles/r04_sep_split_phase_trace.txt).  Unlike round 3's swapped-operand form (lane = row: 32 bytes to each of 32 rows per
// instruction, network 19.2 against 17.0 ms) every instruction here writes whole 128-byte row segments.  Same values, same bytes.  scr: PITCH floats per row,
// 32 rows, private to the wavefront (LDS executes a wavefront's accesses in order: only the compiler has to be kept from reordering them).
template <int BN, bool ADD, int PITCH>
__device__ __forceinline__ void conv_epilogue_lds(f32x16 (&acc)[2][BN / 64], float *__restrict__ Y, const float *__restrict__ scale,
                                                  const float *__restrict__ shift, const float *__restrict__ Add,
                                                  const uint8_t *__restrict__ valid, int m0, int n0, int wm, int wn, int lane, int cout, int relu,
                                                  float post, float *scr) {
    constexpr int NJ = BN / 64;
    const unsigned long long vmask = __ballot(valid[m0 + wm * 64 + lane] != 0);
    const bool all_valid = vmask == ~0ull;                 // wave-uniform
    const float floor_ = relu ? 0.0f : -3.402823466e38f;
    const int colw = n0 + wn * (BN / 2);                   // the wavefront's first column
    const int colb = colw + (lane & 31);
    auto uniform_ptr = [](const void *p) {
        const unsigned long long v = (unsigned long long)p;
        return (void *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v));
    };
    const size_t wbase = (size_t)(m0 + wm * 64) * cout;
    const int wbytes = 64 * cout * 4;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(Y + wbase), 0, wbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<float *>(ADD ? Add + wbase : Y + wbase)), 0, wbytes, 0x00020000);
    const int voff = (4 * (lane >> 5) * cout + colb) * 4;                       // accumulator layout (residual loads)
    const int soff_w = ((lane >> 3) * cout + colw + (lane & 7) * 4) * 4;          // row-piece layout (stores): row lane >> 3 of a group of 8, 16-byte piece lane & 7
    float *wr = scr + 4 * (lane >> 5) * PITCH + (lane & 31);
    const float *rd = scr + (lane >> 3) * PITCH + (lane & 7) * 4;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned eu32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int i = 0; i < 2; i++) {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const float sc = scale[colb + j * 32] * post, sh = shift[colb + j * 32];
            float addv[16];
            if (ADD) {
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const int soff = __builtin_amdgcn_readfirstlane((i * 32 + (q & 3) + 8 * (q >> 2)) * cout * 4);
                    addv[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, voff + j * 128, soff, 0));
                }
            }
#pragma unroll
            for (int q = 0; q < 16; q += 2) {
                const int row = i * 32 + (q & 3) + 8 * (q >> 2);
                f32x2 y = __builtin_elementwise_fma(f32x2{acc[i][j][q], acc[i][j][q + 1]}, f32x2{sc, sc}, f32x2{sh, sh});
                if (ADD) y += f32x2{addv[q], addv[q + 1]};
                float y0 = fmaxf(y[0], floor_), y1 = fmaxf(y[1], floor_);
                if (!all_valid) {
                    const unsigned long long k0 = (((vmask >> row) & 1ull) ? 0x00000000ffffffffull : 0ull) | (((vmask >> (row + 4)) & 1ull) ? 0xffffffff00000000ull : 0ull);
                    const unsigned long long k1 = (((vmask >> (row + 1)) & 1ull) ? 0x00000000ffffffffull : 0ull) | (((vmask >> (row + 5)) & 1ull) ? 0xffffffff00000000ull : 0ull);
                    asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(y0) : "v"(y0), "s"(k0));
                    asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(y1) : "v"(y1), "s"(k1));
                }
                wr[((q & 3) + 8 * (q >> 2)) * PITCH] = y0;
                wr[((q & 3) + 8 * (q >> 2) + 1) * PITCH] = y1;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(rd + 8 * k * PITCH);
                const int soff = __builtin_amdgcn_readfirstlane((i * 32 + 8 * k) * cout * 4);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(eu32x4, v), ry, soff_w + j * 128, soff, 0);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

