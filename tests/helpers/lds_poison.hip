// Test helper (not part of the product): fills the dynamic LDS of every CU with a chosen 32-bit pattern, so that a
// kernel launched afterwards finds inf / NaN bit patterns wherever it reads LDS it has not written itself.
// Built by llm-inference_amd/build.py into tests/helpers/libcmhsa_testhelpers.so; used by tests/test_lds_poison.py.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
__global__ __launch_bounds__(512) void lds_fill_kernel(unsigned pattern, int words, unsigned *sink) {
    extern __shared__ unsigned lds_words[];
    for (int i = threadIdx.x; i < words; i += blockDim.x) lds_words[i] = pattern;
    __syncthreads();
    // read one word back so that the stores cannot be dropped
    if (threadIdx.x == 0 && lds_words[(blockIdx.x * 97) % words] != pattern) sink[0] = 1;
}
}  // namespace

extern "C" int cmhsa_test_fill_lds(unsigned pattern, int bytes_per_workgroup, int workgroups, unsigned *sink_dev, void *stream) {
    if (bytes_per_workgroup <= 0 || bytes_per_workgroup > 160 * 1024 || (bytes_per_workgroup & 3) || workgroups <= 0 || !sink_dev) return -2;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(lds_fill_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       bytes_per_workgroup);
    if (e != hipSuccess) return (int)e;
    int words = bytes_per_workgroup / 4;
    void *args[] = {&pattern, &words, &sink_dev};
    e = hipLaunchKernel(reinterpret_cast<const void *>(lds_fill_kernel), dim3((unsigned)workgroups), dim3(512), args,
                        (size_t)bytes_per_workgroup, static_cast<hipStream_t>(stream));
    return (int)e;
}
