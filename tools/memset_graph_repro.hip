// Does a hipMemsetAsync node captured into a HIP graph replay with its own fill pattern?
// Two small memsets with different patterns and a kernel that adds one to both buffers are
// captured once and replayed; every replay must leave a[i] == 1 and b[i] == 0x01010102.
// Build: hipcc --offload-arch=gfx950 tools/memset_graph_repro.hip -o tools/memset_graph_repro
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 2; } } while (0)
__global__ void bump(unsigned* a, unsigned* b, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { a[i] += 1u; b[i] += 1u; }
}
int main() {
    const int n = 16;  // 64 bytes each, the size of the library's control block
    unsigned *a, *b, ha[16], hb[16];
    hipStream_t s;
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamCreate(&s));
    CK(hipMalloc(&a, 4 * n));
    CK(hipMalloc(&b, 4 * n));
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    CK(hipMemsetAsync(a, 0x00, 4 * n, s));
    CK(hipMemsetAsync(b, 0x01, 4 * n, s));
    hipLaunchKernelGGL(bump, dim3(1), dim3(64), 0, s, a, b, n);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    int bad = 0;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipGraphLaunch(ge, s));
        CK(hipMemcpyAsync(ha, a, 4 * n, hipMemcpyDeviceToHost, s));
        CK(hipMemcpyAsync(hb, b, 4 * n, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
        for (int i = 0; i < n; ++i) bad += (ha[i] != 1u) + (hb[i] != 0x01010102u);
        printf("replay %d: a[0] = %#x (want 0x1)  b[0] = %#x (want 0x1010102)\n", rep, ha[0], hb[0]);
    }
    printf(bad ? "MISMATCH: captured memset nodes do not replay as recorded\n"
               : "ok: captured memset nodes replay as recorded\n");
    return bad ? 1 : 0;
}
