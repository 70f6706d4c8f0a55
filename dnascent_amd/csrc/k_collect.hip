// k_collect.hip -- the bulk result of a batch (dn_collect): what runCNN leaves on a read (detect.cpp:677-731) for every read of
// the batch, compacted on the device so that ONE device-to-host transfer per array brings everything the writer needs.
//
// runCNN reports a position only when its strand 9-mer has 'T' in the middle (detect.cpp:690), for the .detect text and for
// the modbam tags alike; a "call" below is such a position.  Per read the calls keep eventalign's creation order
// (sequencing direction, reads.h:305-372); reads follow each other in batch order.
//
//   kc_npos    positions the CNN has to run on: n_positions of passing reads, 0 otherwise
//   kc_count   calls per read                        (block per read)
//   kc_scan    exclusive scan over the reads -> call_off[n + 1]   (one block)
//   kc_pack    ordered compaction: reference coordinate, query index, reference index, P(EdU) = class 2, P(BrdU) = class 1
//              (detect.cpp:695), the 9-mer as it stands in referenceSeqMappedTo
#include "dn_dev.h"

struct CollectDev {
    const unsigned *coord, *qidx, *ridx; const float *probs;     // eventalign / CNN outputs at ref_off (probs: [3] per position)
    unsigned *cnt; unsigned long long *off;                      // [n], [n + 1]
    unsigned *o_coord, *o_qidx, *o_ridx; float *o_edu, *o_brdu; char *o_kmer;
};

__global__ __launch_bounds__(256) void kc_npos(BatchDev B, unsigned *npos) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < B.n_reads) npos[r] = B.res[r].status == 0 ? B.res[r].n_positions : 0u;
}

__device__ __forceinline__ bool is_call(const BatchDev &B, const CollectDev &C, uint64_t f0, unsigned p) {
    return B.refseq[f0 + C.ridx[f0 + p]] == 'T';          // kmer[4] of the 9-mer centred on the position's reference index
}

__global__ __launch_bounds__(256) void kc_count(BatchDev B, CollectDev C) {
    __shared__ unsigned part[4];
    const int r = blockIdx.x, tid = threadIdx.x;
    const ReadRes &R = B.res[r];
    const unsigned np = R.status == 0 ? R.n_positions : 0u;
    const uint64_t f0 = B.ref_off[r];
    unsigned k = 0;
    for (unsigned p = tid; p < np; p += 256) k += is_call(B, C, f0, p) ? 1u : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) k += __shfl_xor(k, d);
    if ((tid & 63) == 0) part[tid >> 6] = k;
    __syncthreads();
    if (tid == 0) C.cnt[r] = part[0] + part[1] + part[2] + part[3];
}

__global__ __launch_bounds__(256) void kc_scan(BatchDev B, CollectDev C) {
    __shared__ unsigned long long s[256];
    __shared__ unsigned long long base;
    const int tid = threadIdx.x, n = B.n_reads;
    if (tid == 0) base = 0ull;
    __syncthreads();
    for (int r0 = 0; r0 < n; r0 += 256) {
        const int r = r0 + tid;
        const unsigned long long v = r < n ? (unsigned long long)C.cnt[r] : 0ull;
        s[tid] = v;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            const unsigned long long t = tid >= d ? s[tid - d] : 0ull;
            __syncthreads();
            s[tid] += t;
            __syncthreads();
        }
        if (r < n) C.off[r] = base + s[tid] - v;
        __syncthreads();
        if (tid == 255) base += s[255];
        __syncthreads();
    }
    if (tid == 0) C.off[n] = base;
}

__global__ __launch_bounds__(256) void kc_pack(BatchDev B, CollectDev C) {
    __shared__ unsigned part[4];
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const ReadRes &R = B.res[r];
    const unsigned np = R.status == 0 ? R.n_positions : 0u;
    if (np == 0) return;
    const uint64_t f0 = B.ref_off[r];
    unsigned long long out = C.off[r];
    for (unsigned p0 = 0; p0 < np; p0 += 256) {
        const unsigned p = p0 + tid;
        const bool call = p < np && is_call(B, C, f0, p);
        const unsigned long long m = __ballot(call);
        if (lane == 0) part[wave] = (unsigned)__popcll(m);
        __syncthreads();
        unsigned before = 0, total = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) { const unsigned k = part[q]; before += q < wave ? k : 0u; total += k; }
        if (call) {
            const unsigned long long o = out + before + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
            const unsigned ri = C.ridx[f0 + p];
            C.o_coord[o] = C.coord[f0 + p]; C.o_qidx[o] = C.qidx[f0 + p]; C.o_ridx[o] = ri;
            const float *pr = C.probs + (f0 + p) * 3;
            C.o_edu[o] = pr[2]; C.o_brdu[o] = pr[1];                       // detect.cpp:695: class 2 -> EdU, class 1 -> BrdU
            const char *km = B.refseq + f0 + ri - DN_K / 2;
#pragma unroll
            for (int j = 0; j < DN_K; j++) C.o_kmer[o * DN_K + j] = km[j];
        }
        out += total;
        __syncthreads();
    }
}

__global__ void kc_nop() {}
// an empty kernel: a profiling event recorded behind it carries the time at which the stream really got past a preceding
// hipStreamWaitEvent (a bare event record after the wait is stamped early)
void kc_launch_nop(hipStream_t st) { hipLaunchKernelGGL(kc_nop, dim3(1), dim3(1), 0, st); }

void kc_launch_npos(const BatchDev &B, unsigned *npos, hipStream_t st) {
    hipLaunchKernelGGL(kc_npos, dim3((B.n_reads + 255) / 256), dim3(256), 0, st, B, npos);
}
void kc_launch_count(const BatchDev &B, const void *cd, unsigned, hipStream_t st) {
    hipLaunchKernelGGL(kc_count, dim3(B.n_reads), dim3(256), 0, st, B, *reinterpret_cast<const CollectDev *>(cd));
}
void kc_launch_scan(const BatchDev &B, const void *cd, hipStream_t st) {
    hipLaunchKernelGGL(kc_scan, dim3(1), dim3(256), 0, st, B, *reinterpret_cast<const CollectDev *>(cd));
}
void kc_launch_pack(const BatchDev &B, const void *cd, unsigned, hipStream_t st) {
    hipLaunchKernelGGL(kc_pack, dim3(B.n_reads), dim3(256), 0, st, B, *reinterpret_cast<const CollectDev *>(cd));
}
