// Does the row stride of the activation operand limit the K = 95 008 GEMMs?  Each workgroup streams what a 256 x 256 tile of
// gemm_nt_t256 streams from A: per K-tile 256 rows x 128 B, K-tile after K-tile, by LDS-DMA (8 waves, 4 DMAs of 1 KiB each per
// wave and K-tile), nothing else.  layout 0: row-major [M][K] (rows 190 016 B apart); layout 1: K-blocked [K/64][M][64]
// (a K-tile of a row tile is one contiguous 32 KiB block).  240 workgroups like the real launch (12 row tiles x 4 x 5 slices: the
// 4 column tiles of a (slice, row tile) read the SAME bytes).     hipcc --offload-arch=gfx950 -O3 stride_bw.hip -o stride_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) void lds_t;
__global__ __launch_bounds__(512) void stream_kernel(const char* A, long a_bytes, int M, int K, int layout, int splitk, int ncol, float* sink) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[65536];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_m = M / 256;
    const int item = blockIdx.x;                       // z * (ncol * tiles_m) + tn * tiles_m + tm
    const int z = item / (ncol * tiles_m), tm = item % tiles_m;
    const int kts = K / 64;
    const int kt0 = (int)((long)kts * z / splitk), kt1 = (int)((long)kts * (z + 1) / splitk);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(A), 0, (int)(a_bytes > 0x7fffffffL ? 0x7fffffff : a_bytes), 0x00020000);
    const int rl = lane >> 3, dp = lane & 7;
    const long row_b = layout ? 128 : (long)K * 2;
    const long kt_b = layout ? (long)M * 128 : 128;
    uint32_t v[4];
    for (int q = 0; q < 4; ++q) v[q] = (uint32_t)((long)(tm * 256 + wave * 32 + q * 8 + rl) * row_b + dp * 16);
    float acc = 0.f;
    for (int kt = kt0; kt < kt1; ++kt) {
        const uint32_t s = (uint32_t)((long)kt * kt_b);
        const int buf = (kt & 1) * 32768;
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_t*)(smem + buf + (wave * 32 + q * 8) * 128), 16, v[q], s, 0, 0);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // the previous K-tile has landed
        if (kt > kt0) acc += *(float*)(smem + (buf ^ 32768) + tid * 4);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 123.456f) sink[0] = acc;
}
int main(int argc, char** argv) {
    const int M = 3072, K = 95040, splitk = 5, ncol = 4;
    const long bytes = (long)M * K * 2;
    char* A; float* sink;
    hipMalloc(&A, bytes); hipMalloc(&sink, 4); hipMemset(A, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int layout = 0; layout < 2; ++layout)
        for (int nc = 1; nc <= ncol; nc += 3) {
            const int grid = (M / 256) * nc * splitk;
            float best = 1e9;
            for (int rep = 0; rep < 6; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(stream_kernel, dim3(grid), dim3(512), 0, 0, A, bytes, M, K, layout, splitk, nc, sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            printf("layout %d (%s), %d column tiles (%d workgroups): %.1f us, unique %.2f TB/s, requested %.2f TB/s\n", layout,
                   layout ? "K-blocked" : "row-major", nc, grid, best * 1e3, bytes / (best * 1e-3) / 1e12, bytes * (double)nc / (best * 1e-3) / 1e12);
        }
    return 0;
}
