// Probe: does buffer_load ... lds reach LDS addresses >= 64 KiB, or does the M0 base wrap at 16 bits?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ __launch_bounds__(64) void probe(const unsigned* src, unsigned* out, int n_bytes, int base_words) {
    __shared__ __attribute__((aligned(1024))) unsigned sm[24 * 1024];   // 96 KiB
    const int lane = threadIdx.x;
    for (int i = lane; i < 24 * 1024; i += 64) sm[i] = 0xDEADBEEFu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, n_bytes, 0x00020000);
    unsigned char* dst = (unsigned char*)sm + (size_t)base_words * 4;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)dst, 16, lane * 16u, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 24 * 1024; i += 64) out[i] = sm[i];
}
int main() {
    std::vector<unsigned> h(256);
    for (int i = 0; i < 256; ++i) h[i] = 0x5000u + i;
    unsigned *d, *o;
    hipMalloc(&d, 1024); hipMalloc(&o, 96 * 1024);
    hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
    for (int base_kb : {0, 48, 63, 64, 70, 90}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o, 1024, base_kb * 256);
        std::vector<unsigned> r(24 * 1024);
        hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
        int first = -1, count = 0;
        for (int i = 0; i < 24 * 1024; ++i) if (r[i] != 0xDEADBEEFu) { if (first < 0) first = i; ++count; }
        printf("dst base %3d KiB -> first modified word at byte %d (%.1f KiB), %d words modified, value %08x\n", base_kb, first * 4, first * 4 / 1024.0, count, first >= 0 ? r[first] : 0);
    }
    return 0;
}
