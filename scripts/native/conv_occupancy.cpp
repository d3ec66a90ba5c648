// Diagnostic: occupancy of the conv kernels as the runtime sees it.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o conv_occupancy conv_occupancy.cpp
#include "../../deep-q-learning_tron_amd/csrc/tron_conv.hip"
#include <stdio.h>
template <int S, bool CODES>
void report(const char *name)
{
    using C = Cfg<S>;
    auto kern = k_conv3x3<S, CODES>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
    int nb = -1;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, THREADS, C::LDS_BYTES);
    hipFuncAttributes a;
    hipFuncGetAttributes(&a, reinterpret_cast<const void *>(kern));
    printf("%s: LDS %zu B, blocks/CU %d (err %d), regs %d, static LDS %zu, maxDyn %d\n", name, C::LDS_BYTES, nb, (int)e, a.numRegs,
           a.sharedSizeBytes, a.maxDynamicSharedSizeBytes);
}
int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s: LDS per CU %zu, per block %zu, regs/block %d\n", p.name, p.maxSharedMemoryPerMultiProcessor, p.sharedMemPerBlock, p.regsPerBlock);
    report<12, false>("conv S=12");
    report<12, true>("conv S=12 codes");
    report<26, false>("conv S=26");
    report<26, true>("conv S=26 codes");
    return 0;
}
