// hbm_read.hip -- what a cold READ stream can reach on this chip, independent of any library kernel: 16-byte loads over a 4 GB
// buffer (16 x the Infinity Cache), U loads in flight per lane, interleaved (grid-stride) or block-contiguous chunks, several grid
// sizes.  hipcc --offload-arch=gfx950 -O3 scripts/experiments/hbm_read.hip -o /tmp/hbm_read && /tmp/hbm_read
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_fill(u32x4* q, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const uint32_t h = (uint32_t)i * 2654435761u;
        q[i] = u32x4{h, h ^ 0x9e3779b9u, h * 3u, ~h};
    }
}

template <int U, bool CONTIG>
__global__ void __launch_bounds__(256) k_read(const u32x4* __restrict__ p, uint32_t* __restrict__ sink, int64_t n16) {
    const int64_t nthreads = (int64_t)gridDim.x * 256;
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
    if (CONTIG) {                                   // every block walks its own contiguous chunk
        const int64_t per_block = n16 / gridDim.x;
        const u32x4* q = p + (int64_t)blockIdx.x * per_block;
        for (int64_t i = threadIdx.x; i + (U - 1) * 256 < per_block; i += U * 256) {
            u32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = q[i + u * 256];
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
        }
    } else {
        for (int64_t i = tid; i + (U - 1) * nthreads < n16; i += U * nthreads) {
            u32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = p[i + u * nthreads];
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int U, bool CONTIG>
static void run(const u32x4* p, uint32_t* sink, int64_t n16, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_read<U, CONTIG>), dim3(blocks), dim3(256), 0, 0, p, sink, n16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k_read<U, CONTIG>), dim3(blocks), dim3(256), 0, 0, p, sink, n16);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("U %2d %-10s blocks %6d: %7.1f us  %6.0f GB/s\n", U, CONTIG ? "contiguous" : "interleaved", blocks, ms / 3 * 1e3, (double)n16 * 16 / (ms / 3 * 1e-3) / 1e9);
}

// short cold streams: every launch reads the next `mb` megabytes of the buffer (no launch sees bytes an earlier one left in a cache)
template <int U>
static void run_slices(const u32x4* p, uint32_t* sink, int64_t total16, double mb, int blocks) {
    const int64_t n16 = (int64_t)(mb * 1e6 / 16) / (U * 256 * (int64_t)blocks) * (U * 256 * (int64_t)blocks);
    if (n16 == 0) return;                                       // slice smaller than one pass of this grid
    const int nl = (int)(total16 / n16) < 30 ? (int)(total16 / n16) : 30;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < nl; ++r) hipLaunchKernelGGL((k_read<U, false>), dim3(blocks), dim3(256), 0, 0, p + (int64_t)r * n16, sink, n16);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("slices of %6.1f MB, U %d, blocks %5d: %6.1f us per launch  %6.0f GB/s (%d back-to-back launches)\n", (double)n16 * 16 / 1e6, U, blocks, ms / nl * 1e3,
           (double)n16 * 16 / (ms / nl * 1e-3) / 1e9, nl);
}

int main() {
    const int64_t bytes = 4ll << 30, n16 = bytes / 16;
    u32x4* p; uint32_t* sink;
    hipMalloc(&p, bytes); hipMalloc(&sink, 4);
    hipMemset(p, 1, bytes);
    hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, p, n16);          // not a constant pattern: a cheap hash of the index
    hipDeviceSynchronize();
    // how much one CU can pull: fewer blocks than CUs (one block = 4 waves on one CU)
    for (int blocks : {32, 64, 128, 192}) { run<8, false>(p, sink, n16 / 4, blocks); run<16, true>(p, sink, n16 / 4, blocks); }
    for (double mb : {19.0, 52.0, 104.0, 416.0})
        for (int blocks : {256, 512, 1024, 2048}) { run_slices<4>(p, sink, n16, mb, blocks); run_slices<8>(p, sink, n16, mb, blocks); }
    for (int blocks : {256, 512, 1024, 2048, 4096, 8192, 65536}) {
        run<4, false>(p, sink, n16, blocks);
        run<8, false>(p, sink, n16, blocks);
        run<8, true>(p, sink, n16, blocks);
        run<16, true>(p, sink, n16, blocks);
    }
    return 0;
}
