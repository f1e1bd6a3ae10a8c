#include <hip/hip_runtime.h>
#include <stdint.h>
template <int CTRL, int RM = 0xF, int BM = 0xF>
__device__ __forceinline__ uint32_t dpp(uint32_t old, uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)x, CTRL, RM, BM, false); }
__device__ __forceinline__ uint32_t scan(uint32_t v) {
    v += dpp<0x111>(0, v); v += dpp<0x112>(0, v); v += dpp<0x114>(0, v); v += dpp<0x118>(0, v);
    v += dpp<0x142, 0xA>(0, v); v += dpp<0x143, 0xC>(0, v);
    return v;
}
__global__ void k(uint32_t* o, const uint32_t* in, uint32_t carry) {
    uint32_t x = in[threadIdx.x];
    uint32_t a = dpp<0x138>(carry, x);
    o[threadIdx.x] = a;
    o[64 + threadIdx.x] = scan(x);
}
int main() {
    uint32_t *d, *i; hipMalloc(&d, 512); hipMalloc(&i, 256);
    uint32_t h[64]; for (int j = 0; j < 64; j++) h[j] = j * 3 + 1;
    hipMemcpy(i, h, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, i, 777u);
    uint32_t r[128]; hipMemcpy(r, d, 512, hipMemcpyDeviceToHost);
    int bad = 0; uint32_t s = 0;
    for (int j = 0; j < 64; j++) { s += h[j]; if (r[j] != (j ? h[j - 1] : 777u)) bad++; if (r[64 + j] != s) bad++; }
    printf("bad=%d r0=%u r1=%u scan63=%u expect %u\n", bad, r[0], r[1], r[127], s);
    return bad != 0;
}
