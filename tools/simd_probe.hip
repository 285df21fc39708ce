// Where do the waves of a workgroup run?  Prints (wave, SIMD, CU) of one block of 512 and of 768 threads from
// HW_REG_HW_ID (gfx9: wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13]).
// The split-half reader (csrc/plsx_splitfused.h) assigns its wave roles by this map.
//   hipcc --offload-arch=gfx950 -O2 tools/simd_probe.hip -o tools/bin/simd_probe && tools/bin/simd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_probe(unsigned* out)
{
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id;
}

int main()
{
    unsigned* d;
    hipMalloc(&d, 64 * 16 * 4);
    for (int threads : {512, 768, 256, 1024}) {
        hipMemset(d, 0xff, 64 * 16 * 4);
        hipLaunchKernelGGL(k_probe, dim3(3), dim3(threads), 0, 0, d);
        hipDeviceSynchronize();
        std::vector<unsigned> h(64 * 16);
        hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
        for (int b = 0; b < 3; ++b) {
            printf("threads %4d block %d:", threads, b);
            for (int w = 0; w < threads / 64; ++w) {
                const unsigned id = h[b * 16 + w];
                printf("  w%d:simd%u/cu%u/se%u", w, (id >> 4) & 3, (id >> 8) & 15, (id >> 13) & 7);
            }
            printf("\n");
        }
    }
    return 0;
}
