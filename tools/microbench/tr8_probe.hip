// Semantics probe (development tool): what ds_read_b64_tr_b8 delivers to each lane, and where global_load_lds_dwordx4
// puts each lane's 16 bytes.  Prints the lane/byte permutation so the resample kernel's LDS image can be laid out for it.
//   hipcc --offload-arch=gfx950 -O3 tr8_probe.hip -o tr8_probe && ./tr8_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef int v2i __attribute__((ext_vector_type(2)));

// every lane owns 8 bytes at lds[lane * 8 ..]; byte value = (lane & 15) * 8 + byte index, so the result names its source
__global__ void tr8(uint32_t *out)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[64 * 8];
    for (int b = 0; b < 8; ++b) lds[threadIdx.x * 8 + b] = (uint8_t)((threadIdx.x & 15) * 8 + b);
    __syncthreads();
    v2i v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i __attribute__((address_space(3))) *)(lds + threadIdx.x * 8));
    out[threadIdx.x * 2] = (uint32_t)v[0];
    out[threadIdx.x * 2 + 1] = (uint32_t)v[1];
}

__global__ void glds(const uint32_t *src, uint32_t *out)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    // lane l reads 16 B at src + 64 * l bytes (a stride, to tell lane order from address order)
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + threadIdx.x * 16), (void __attribute__((address_space(3))) *)(lds + 64), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}

int main()
{
    uint32_t *d;
    CK(hipMalloc(&d, 1 << 16));
    tr8<<<1, 64>>>(d);
    std::vector<uint32_t> h(128);
    CK(hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost));
    printf("ds_read_b64_tr_b8: lane -> 8 x (source lane-in-group . source byte)\n");
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 8; ++j) { const unsigned b = (h[l * 2 + j / 4] >> (8 * (j & 3))) & 0xff; printf(" %2u.%u", b >> 3, b & 7); }
        printf("\n");
        if (l == 15) l = 47; // groups behave alike: print the first and the last
    }
    std::vector<uint32_t> s(64 * 16);
    for (size_t i = 0; i < s.size(); ++i) s[i] = (uint32_t)i;
    uint32_t *ds;
    CK(hipMalloc(&ds, s.size() * 4));
    CK(hipMemcpy(ds, s.data(), s.size() * 4, hipMemcpyHostToDevice));
    glds<<<1, 64>>>(ds, d);
    std::vector<uint32_t> o(1024);
    CK(hipMemcpy(o.data(), d, 4096, hipMemcpyDeviceToHost));
    printf("global_load_lds_dwordx4 (lane l reads src dwords 16l..16l+3, LDS base = dword 64):\n");
    for (int i = 56; i < 64 + 64 * 4 + 8; ++i) {
        if (o[i] == 0xdeadbeefu) { if (i < 64 || i >= 64 + 256) printf(" [%d untouched]", i); continue; }
        if ((i - 64) % 16 == 0) printf("\n lds dword %3d:", i);
        printf(" %u", o[i]);
    }
    printf("\n");
    return 0;
}
