// TEST-ONLY stand-in for the HIP runtime entry points libgmmhmm's HOST code calls (tests/test_host_sanitized.py): the host
// halves of speech-recognition_amd/csrc/*.hip are compiled with -fsanitize=address,undefined (hipcc --cuda-host-only, the
// product sources as they are, no #ifdef) and linked against THIS file instead of libamdhip64, so that the plan builders,
// chunk planners, upload layouts, transcript expansion and session set-up run under ASan/UBSan on a machine without a GPU.
//   * "device" memory is host memory (zero-filled: what a kernel would have written reads back as 0); every hipMemcpy is a
//     memcpy, so a copy that overruns either side is an ASan report with the product's stack;
//   * kernel launches are accepted and do nothing; streams and events are opaque tokens; one device, 256 CUs, 288 GB.
// Nothing in the product links, includes or loads this file.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

namespace {
struct call_config { dim3 grid, block; size_t shmem; hipStream_t stream; };
thread_local call_config g_cfg[8];
thread_local int g_depth = 0;
hipError_t g_last = hipSuccess;
long g_launches = 0;
}

extern "C" {

long hipstub_launches() { return g_launches; }

hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidDevice; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipGetLastError() { hipError_t e = g_last; g_last = hipSuccess; return e; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : (e == hipErrorOutOfMemory ? "out of memory (stub)" : "error (stub)"); }

hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600* p, int) {
    memset(p, 0, sizeof *p);
    strcpy(p->name, "hipstub gfx950");
    p->multiProcessorCount = 256;
    p->totalGlobalMem = (size_t)288 << 30;
    p->sharedMemPerBlock = 160 * 1024;
    p->maxThreadsPerBlock = 1024;
    p->warpSize = 64;
    p->clockRate = 2400000;
    return hipSuccess;
}
hipError_t hipMemGetInfo(size_t* free_b, size_t* total) { *free_b = (size_t)64 << 30; *total = (size_t)288 << 30; return hipSuccess; }

hipError_t hipMalloc(void** p, size_t n) {
    if (n > ((size_t)8 << 30)) { *p = nullptr; return hipErrorOutOfMemory; }   // (a size that only a wrong plan asks for)
    *p = calloc(n ? n : 1, 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = calloc(n ? n : 1, 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostRegister(void*, size_t, unsigned) { return hipSuccess; }
hipError_t hipHostUnregister(void*) { return hipSuccess; }

hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { if (n) memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t) {
    for (size_t r = 0; r < h; ++r) if (w) memcpy((char*)d + r * dp, (const char*)s + r * sp, w);
    return hipSuccess;
}
hipError_t hipMemset(void* d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { if (n) memset(d, v, n); return hipSuccess; }

hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)malloc(8); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free((void*)s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { free((void*)e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.001f; return hipSuccess; }

hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void*, int, size_t) { *n = 4; return hipSuccess; }
hipError_t hipLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t) { ++g_launches; return hipSuccess; }

// what clang's host side of a <<<>>> launch and of a translation unit with kernels calls
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t stream) {
    if (g_depth < 8) g_cfg[g_depth] = call_config{grid, block, shmem, stream};
    ++g_depth;
    return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3* grid, dim3* block, size_t* shmem, hipStream_t* stream) {
    if (g_depth > 0) --g_depth;
    const call_config& c = g_cfg[g_depth < 8 ? g_depth : 7];
    *grid = c.grid; *block = c.block; *shmem = c.shmem; *stream = c.stream;
    return hipSuccess;
}
void** __hipRegisterFatBinary(const void*) { static void* handle[2]; return handle; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}

}  // extern "C"
