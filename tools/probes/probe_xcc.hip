// Which XCD does workgroup i of a 1-D / 2-D grid land on?  (s_getreg HW_REG_XCC_ID, gfx942/gfx950.)  The fused scoring kernel
// wants the workgroups of one (batch, kv head) unit on ONE XCD, so that their hand-offs can go through that XCD's L2.
// build: hipcc --offload-arch=gfx950 -O3 -o probe_xcc probe_xcc.hip ; run: ./probe_xcc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(int *out)
{
    if (threadIdx.x == 0) {
        const int id = blockIdx.y * gridDim.x + blockIdx.x;
        out[id] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15;       // XCC_ID, bits 3:0
    }
    // keep the workgroup alive for a while so that the whole grid is resident at once
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000) { }
}
int main()
{
    int *d;
    hipMalloc(&d, 4096 * sizeof(int));
    for (int trial = 0; trial < 3; ++trial) {
        const dim3 grids[3] = {dim3(512), dim3(128), dim3(256, 2)};
        const dim3 g = grids[trial];
        const int n = g.x * g.y;
        hipMemset(d, 0xff, 4096 * sizeof(int));
        hipLaunchKernelGGL(k, g, dim3(256), 0, 0, d);
        std::vector<int> h(n);
        hipMemcpy(h.data(), d, n * sizeof(int), hipMemcpyDeviceToHost);
        int bad = 0, hist[16] = {0};
        for (int i = 0; i < n; ++i) { hist[h[i] & 15]++; if (h[i] != i % 8) ++bad; }
        printf("grid (%u,%u): workgroups not on XCC (id %% 8): %d of %d; first 24:", g.x, g.y, bad, n);
        for (int i = 0; i < 24; ++i) printf(" %d", h[i]);
        printf("; per XCC:");
        for (int i = 0; i < 8; ++i) printf(" %d", hist[i]);
        printf("\n");
    }
    return 0;
}
