// The host pipeline's pattern of streams and events without the codec: chunks that are copied in (small), worked on
// (a short kernel) and copied out (two large copies and two small ones), four sets of buffers, the host waiting for
// the chunk four back to have drained before it enqueues the next.  Prints how long the host waited per chunk.
// usage: pipepattern [MiB per large copy] [chunks] [variant]
//   variant 0: as the library does   1: no small copies out   2: events with timing enabled
//   3: small copies out first        4: copy-out stream of normal priority
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void touch(uint64_t* p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] += 1;
}

int main(int argc, char** argv)
{
    const size_t mib = argc > 1 ? atoi(argv[1]) : 128;
    const int n = argc > 2 ? atoi(argv[2]) : 17;
    const int variant = argc > 3 ? atoi(argv[3]) : 0;
    const size_t bytes = mib << 20, small = 1 << 20;
    constexpr int NSTAGE = 4;
    char* dev[NSTAGE][2]; char* dsmall[NSTAGE]; char* host[2]; char* hsmall; char* hin;
    for (int b = 0; b < NSTAGE; b++) { CK(hipMalloc(&dev[b][0], bytes)); CK(hipMalloc(&dev[b][1], bytes)); CK(hipMalloc(&dsmall[b], 4 * small)); }
    CK(hipHostMalloc(&host[0], bytes * (size_t)n)); CK(hipHostMalloc(&host[1], bytes * (size_t)n));
    CK(hipHostMalloc(&hsmall, 2 * small * (size_t)n)); CK(hipHostMalloc(&hin, 2 * small));
    int least = 0, greatest = 0;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t copy_in, work, copy_out;
    CK(hipStreamCreateWithPriority(&copy_in, hipStreamNonBlocking, greatest));
    CK(hipStreamCreateWithPriority(&copy_out, hipStreamNonBlocking, variant == 4 ? 0 : greatest));
    CK(hipStreamCreateWithFlags(&work, hipStreamNonBlocking));
    hipEvent_t copied_in[NSTAGE], coded[NSTAGE], drained[NSTAGE];
    const unsigned flags = variant == 2 ? hipEventDefault : hipEventDisableTiming;
    for (int b = 0; b < NSTAGE; b++) { CK(hipEventCreateWithFlags(&copied_in[b], flags)); CK(hipEventCreateWithFlags(&coded[b], flags)); CK(hipEventCreateWithFlags(&drained[b], flags)); }
    for (int rep = 0; rep < 3; rep++) {
        std::vector<double> waited(n, 0.0);
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < n; k++) {
            const int b = k % NSTAGE;
            const auto ta = std::chrono::steady_clock::now();
            if (k >= NSTAGE) CK(hipEventSynchronize(drained[b]));
            waited[k] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ta).count();
            CK(hipMemcpyAsync(dsmall[b], hin, small, hipMemcpyHostToDevice, copy_in));
            CK(hipMemcpyAsync(dsmall[b] + small, hin + small, small, hipMemcpyHostToDevice, copy_in));
            CK(hipEventRecord(copied_in[b], copy_in));
            CK(hipStreamWaitEvent(work, copied_in[b], 0));
            CK(hipMemsetAsync(dsmall[b] + 2 * small, 0, 144, work));
            hipLaunchKernelGGL(touch, dim3(2048), dim3(256), 0, work, reinterpret_cast<uint64_t*>(dev[b][0]), bytes / 8);
            hipLaunchKernelGGL(touch, dim3(2048), dim3(256), 0, work, reinterpret_cast<uint64_t*>(dev[b][1]), bytes / 8);
            CK(hipEventRecord(coded[b], work));
            CK(hipStreamWaitEvent(copy_out, coded[b], 0));
            if (variant == 3) {
                CK(hipMemcpyAsync(hsmall + 2 * small * (size_t)k, dsmall[b] + 2 * small, small, hipMemcpyDeviceToHost, copy_out));
                CK(hipMemcpyAsync(hsmall + 2 * small * (size_t)k + small, dsmall[b] + 3 * small, small / 2, hipMemcpyDeviceToHost, copy_out));
            }
            CK(hipMemcpyAsync(host[0] + bytes * (size_t)k, dev[b][0], bytes, hipMemcpyDeviceToHost, copy_out));
            CK(hipMemcpyAsync(host[1] + bytes * (size_t)k, dev[b][1], bytes, hipMemcpyDeviceToHost, copy_out));
            if (variant != 1 && variant != 3) {
                CK(hipMemcpyAsync(hsmall + 2 * small * (size_t)k, dsmall[b] + 2 * small, small, hipMemcpyDeviceToHost, copy_out));
                CK(hipMemcpyAsync(hsmall + 2 * small * (size_t)k + small, dsmall[b] + 3 * small, small / 2, hipMemcpyDeviceToHost, copy_out));
            }
            CK(hipEventRecord(drained[b], copy_out));
        }
        CK(hipStreamSynchronize(work)); CK(hipStreamSynchronize(copy_out));
        const double total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (rep == 0) continue;
        printf("variant %d, %zu MiB x 2 a chunk, %d chunks: %.1f ms, %.1f GB/s; host waited per chunk:", variant, mib, n, total, 2.0 * bytes * n / total / 1e6);
        for (int k = 0; k < n; k++) printf(" %.1f", waited[k]);
        printf("\n");
    }
    return 0;
}
