// experiment: what residency does the runtime report for k_render?
#include <cstdio>
#include "pcr_kernels.hip.h"
int main() {
    int n = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pcr::k_render<0>, 1024, 0);
    printf("k_render<0> blocks/CU by API: %d (%s)\n", n, hipGetErrorString(e));
    hipFuncAttributes at;
    hipFuncGetAttributes(&at, (const void*)pcr::k_render<0>);
    printf("sharedSizeBytes %zu numRegs %d maxThreads %d\n", at.sharedSizeBytes, at.numRegs, at.maxThreadsPerBlock);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("sharedMemPerMultiprocessor %zu sharedMemPerBlock %zu maxThreadsPerMP %d regsPerMP %d\n", p.sharedMemPerMultiprocessor, p.sharedMemPerBlock, p.maxThreadsPerMultiProcessor, p.regsPerMultiprocessor);
    return 0;
}
