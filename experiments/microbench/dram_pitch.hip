// Micro-benchmark (round 5): does HBM read bandwidth depend on how much of a row one "step" of a GEMM-like kernel fetches?
// Every workgroup owns 256 consecutive rows of a [rows][pitch bytes] matrix and reads them in steps of `seg` bytes per row
// (seg = 128 is what conv1x1_gemm_kernel fetches per K step), all columns, summing into a register (no LDS, no math).
//   hipcc --offload-arch=gfx950 -O3 dram_pitch.hip -o dram_pitch && ./dram_pitch
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int SEG>      // bytes of one row fetched per step (128, 256, 512, 1024)
__global__ __launch_bounds__(256) void walk(const uint4* __restrict__ m, int pitch16, unsigned* out) {
    constexpr int LPR = SEG / 16;                 // lanes per row in a step
    constexpr int RPS = 256 / LPR;                // rows per step
    const size_t row0 = (size_t)blockIdx.x * 256;
    unsigned acc = 0;
    for (int c0 = 0; c0 < pitch16; c0 += LPR)                       // column steps (K steps)
        for (int r0 = 0; r0 < 256; r0 += RPS) {                     // the 256 rows of the tile
            const int r = r0 + threadIdx.x / LPR, c = c0 + threadIdx.x % LPR;
            const uint4 v = m[(row0 + r) * pitch16 + c];
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
    if (acc == 0x12345678u) out[0] = acc;
}
// the same bytes in the order a GEMM K loop issues them: per step ALL 256 rows x SEG bytes (rows outer inside a step)
template <int SEG>
__global__ __launch_bounds__(256) void walk_sequential(const uint4* __restrict__ m, int pitch16, unsigned* out) {
    const size_t base = (size_t)blockIdx.x * 256 * pitch16;
    unsigned acc = 0;
    for (size_t i = threadIdx.x; i < (size_t)256 * pitch16; i += 256) {
        const uint4 v = m[base + i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <typename K> float time_it(K k, const uint4* m, int pitch16, unsigned* out, int wgs) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, m, pitch16, out);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, m, pitch16, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 10;
}

int main() {
    const size_t bytes = (size_t)512 << 20;       // 512 MB: beyond the 256 MB Infinity Cache
    uint4* m; unsigned* out;
    hipMalloc(&m, bytes); hipMalloc(&out, 4);
    hipMemset(m, 1, bytes);
    for (int pitch : {128, 512, 1024, 2048}) {
        const int pitch16 = pitch / 16;
        const int wgs = (int)(bytes / ((size_t)256 * pitch));
        printf("row pitch %4d B (%d workgroups x 256 rows): ", pitch, wgs);
        printf("sequential %.2f TB/s", bytes / time_it(walk_sequential<128>, m, pitch16, out, wgs) / 1e9);
        printf(" | 128 B per row and step %.2f", bytes / time_it(walk<128>, m, pitch16, out, wgs) / 1e9);
        if (pitch >= 256) printf(" | 256 B %.2f", bytes / time_it(walk<256>, m, pitch16, out, wgs) / 1e9);
        if (pitch >= 512) printf(" | 512 B %.2f", bytes / time_it(walk<512>, m, pitch16, out, wgs) / 1e9);
        if (pitch >= 1024) printf(" | 1024 B %.2f", bytes / time_it(walk<1024>, m, pitch16, out, wgs) / 1e9);
        printf(" TB/s\n");
    }
    return 0;
}
