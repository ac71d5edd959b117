// rowsel.hpp — the ascending list of the rows a predicate holds for, built on the device (round 6).
// Three small launches (no host sync, no memset node: a captured step stays one hipGraph): per 1024 rows a count, one
// workgroup's exclusive scan of the counts (+ the total), then every row's rank = its block's offset + the rows before it
// inside the block (wave ballots).  Ascending order makes whatever is summed over the list a function of the data alone:
// the same bits run to run.  Users: nerf_bwd.hip (the points with a gradient), nerf_geom.hip (the samples with a density).
//
// Pred: a trivially copyable functor, `bool operator()(long long row) const` and `void visit(long long row, bool on) const`
// (called once per row by the writing pass: the geometry path stores the unlisted rows' output there).
#pragma once
#include <hip/hip_runtime.h>

namespace nfx {
namespace rowsel {

constexpr int kBlockRows = 1024;

template <class Pred>
__global__ __launch_bounds__(256) void count_kernel(Pred pred, long long n, int* __restrict__ block_count) {
    __shared__ int s[4];
    const int tid = threadIdx.x;
    int c = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long long row = (long long)blockIdx.x * kBlockRows + r * 256 + tid;
        c += __popcll(__ballot(row < n && pred(row)));
    }
    if ((tid & 63) == 0) s[tid >> 6] = c;
    __syncthreads();
    if (tid == 0) block_count[blockIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
}

template <class Pred>
__global__ __launch_bounds__(256) void write_kernel(Pred pred, long long n, const int* __restrict__ block_offset,
                                                    int* __restrict__ list) {
    __shared__ int s[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    bool f[4];
    unsigned long long m[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long long row = (long long)blockIdx.x * kBlockRows + r * 256 + tid;
        f[r] = row < n && pred(row);
        if (row < n) pred.visit(row, f[r]);
        m[r] = __ballot(f[r]);
        if (lane == 0) s[r * 4 + wave] = __popcll(m[r]);
    }
    __syncthreads();
    int before = block_offset[blockIdx.x];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k == wave && f[r])
                list[before + __popcll(m[r] & ((1ull << lane) - 1ull))] = (int)((long long)blockIdx.x * kBlockRows + r * 256 + tid);
            before += s[r * 4 + k];
        }
    }
}

// workspace: [count, 3 pad][one count per 1024 rows, padded to 4][n rows of indices] (32-bit words)
inline long long blocks(long long n) { return (n + kBlockRows - 1) / kBlockRows; }
inline size_t workspace_bytes(long long n) {
    return n <= 0 ? 0 : (size_t)(4 + (blocks(n) + 3) / 4 * 4 + (n + 3) / 4 * 4) * sizeof(int);
}
inline int* count_of(void* ws) { return static_cast<int*>(ws); }
inline int* list_of(void* ws, long long n) { return static_cast<int*>(ws) + 4 + (blocks(n) + 3) / 4 * 4; }

// exclusive scan of the block counts in place (one workgroup, 1024 counts per pass) and the total: nerf_bwd.hip
int launch_scan(int* block_count, int n_blocks, int* count, hipStream_t st);

// list_of(ws, n)[0 .. *count_of(ws)) = the rows of [0, n) with pred(row), ascending; n < 2^31
template <class Pred>
int build(Pred pred, long long n, void* ws, hipStream_t st) {
    const long long nb = blocks(n);
    int* block_count = count_of(ws) + 4;
    hipLaunchKernelGGL(count_kernel<Pred>, dim3((unsigned)nb), dim3(256), 0, st, pred, n, block_count);
    int rc = launch_scan(block_count, (int)nb, count_of(ws), st);
    if (rc) return rc;
    hipLaunchKernelGGL(write_kernel<Pred>, dim3((unsigned)nb), dim3(256), 0, st, pred, n, (const int*)block_count, list_of(ws, n));
    return (int)hipGetLastError();
}

}  // namespace rowsel
}  // namespace nfx
