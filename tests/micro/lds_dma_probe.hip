// Probe: buffer_load_dwordx4 ... lds (LDS-DMA) semantics on gfx950.
//  (1) destination is M0-base + lane*16 (lane-linear)?  (2) do out-of-range lanes write ZEROS or skip the write?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef int v4i32 __attribute__((ext_vector_type(4)));
__global__ void probe(const unsigned* src, unsigned* out, int n_bytes) {
    __shared__ __attribute__((aligned(16))) unsigned sm[64 * 4 * 2];
    const int lane = threadIdx.x;
    for (int i = lane; i < 64 * 4 * 2; i += 64) sm[i] = 0xDEADBEEFu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, n_bytes, 0x00020000);
    // lanes with (lane % 5 == 4) point out of range; others read chunk (63 - lane) -> tests per-lane source
    unsigned off = (lane % 5 == 4) ? 0x7FFFFFF0u : (unsigned)(63 - lane) * 16u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)sm, 16, off, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);   // vmcnt(0) lgkmcnt(0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 64 * 4 * 2; i += 64) out[i] = sm[i];
}
int main() {
    std::vector<unsigned> h(64 * 4);
    for (int i = 0; i < 64 * 4; ++i) h[i] = 0x1000u + i;
    unsigned *d, *o;
    hipMalloc(&d, h.size() * 4); hipMalloc(&o, 64 * 4 * 2 * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o, (int)(h.size() * 4));
    std::vector<unsigned> r(64 * 4 * 2);
    if (hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("FAILED\n"); return 1; }
    int linear_ok = 1, oob_zero = 1, oob_skipped = 1, tail_untouched = 1;
    for (int l = 0; l < 64; ++l) {
        for (int k = 0; k < 4; ++k) {
            unsigned v = r[l * 4 + k];
            if (l % 5 == 4) { if (v != 0) oob_zero = 0; if (v != 0xDEADBEEFu) oob_skipped = 0; }
            else if (v != 0x1000u + (63 - l) * 4 + k) linear_ok = 0;
        }
    }
    for (int i = 256; i < 512; ++i) if (r[i] != 0xDEADBEEFu) tail_untouched = 0;
    printf("lane_linear=%d oob_writes_zero=%d oob_skips_write=%d tail_untouched=%d  first words: %08x %08x %08x %08x | lane4: %08x\n",
           linear_ok, oob_zero, oob_skipped, tail_untouched, r[0], r[1], r[2], r[3], r[16]);
    return 0;
}
