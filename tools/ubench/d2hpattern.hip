// Copies out of device memory into pinned host memory, one after the other, each timed: does a long series of
// hipMemcpyAsync on one stream keep its rate?  Variants: one stream; two streams taking turns; one stream with a
// hipStreamSynchronize every eighth copy; one stream with an event wait on the host every fourth copy (what a pipeline
// that recycles its buffers does).
// usage: d2hpattern [MiB per copy] [copies]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv)
{
    const size_t mib = argc > 1 ? atoi(argv[1]) : 128;
    const int n = argc > 2 ? atoi(argv[2]) : 48;
    const size_t bytes = mib << 20;
    char* dev = nullptr; char* host = nullptr;
    CK(hipMalloc(&dev, bytes * 4));
    CK(hipHostMalloc(&host, bytes * (size_t)n, hipHostMallocDefault));
    CK(hipMemset(dev, 1, bytes * 4));
    int least = 0, greatest = 0;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    hipStream_t s[2];
    CK(hipStreamCreateWithPriority(&s[0], hipStreamNonBlocking, greatest));
    CK(hipStreamCreateWithPriority(&s[1], hipStreamNonBlocking, greatest));
    std::vector<hipEvent_t> a(n), b(n);
    for (int i = 0; i < n; i++) { CK(hipEventCreate(&a[i])); CK(hipEventCreate(&b[i])); }
    const char* names[4] = {"one stream", "two streams in turn", "one stream, synchronised every 8th copy", "one stream, host waits for copy i-4 before copy i"};
    for (int variant = 0; variant < 4; variant++) {
        for (int rep = 0; rep < 2; rep++) {
            for (int i = 0; i < n; i++) {
                hipStream_t q = variant == 1 ? s[i & 1] : s[0];
                if (variant == 3 && i >= 4) CK(hipEventSynchronize(b[i - 4]));
                CK(hipEventRecord(a[i], q));
                CK(hipMemcpyAsync(host + bytes * (size_t)i, dev + bytes * (size_t)(i & 3), bytes, hipMemcpyDeviceToHost, q));
                CK(hipEventRecord(b[i], q));
                if (variant == 2 && (i & 7) == 7) CK(hipStreamSynchronize(q));
            }
            CK(hipStreamSynchronize(s[0])); CK(hipStreamSynchronize(s[1]));
            if (rep == 0) continue;
            printf("%s, %zu MiB a copy, ms per copy:", names[variant], mib);
            float total = 0;
            for (int i = 0; i < n; i++) { float ms = 0; CK(hipEventElapsedTime(&ms, a[i], b[i])); total += ms; printf(" %.1f", ms); }
            float span = 0; CK(hipEventElapsedTime(&span, a[0], b[n - 1]));
            printf("\n   first to last %.1f ms, %.1f GB/s\n", span, (double)bytes * n / span / 1e6);
        }
    }
    return 0;
}
