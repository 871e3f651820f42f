// ipc_probe — can one process's kernel read another process's hipMalloc'ed buffer in place (hipIpcGetMemHandle /
// hipIpcOpenMemHandle), which is what reading a neighbour's halo rows over xGMI without an exchange step would need?
// Two processes on ONE GPU (the fork happens before either touches HIP).  Prints what worked and what a dependent
// read costs.   hipcc --offload-arch=gfx950 -O2 tools/ipc_probe.hip -o tools/ipc_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <unistd.h>

#define CHECK(x)                                                                         \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            printf("[%s] %s failed: %s\n", who, #x, hipGetErrorString(e_));              \
            fflush(stdout);                                                              \
            _exit(3);                                                                    \
        }                                                                                \
    } while (0)

__global__ void k_read(const uint32_t* src, uint32_t* dst, size_t n)
{
    for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i] + 1u;
}

int main()
{
    int to_child[2], to_parent[2];
    if (pipe(to_child) || pipe(to_parent)) return 2;
    const size_t n = 6000 * 4 / 2;  // two u16 rows of 6000 samples, as dwords... (4 rows worth)
    pid_t pid = fork();
    if (pid == 0) {  // child: the reader
        const char* who = "reader";
        hipIpcMemHandle_t h;
        if (read(to_child[0], &h, sizeof(h)) != (ssize_t)sizeof(h)) _exit(4);
        CHECK(hipSetDevice(0));
        void* peer = nullptr;
        CHECK(hipIpcOpenMemHandle(&peer, h, hipIpcMemLazyEnablePeerAccess));
        uint32_t* dst;
        CHECK(hipMalloc(&dst, n * 4));
        hipEvent_t a, b;
        CHECK(hipEventCreate(&a));
        CHECK(hipEventCreate(&b));
        for (int it = 0; it < 3; ++it) {
            CHECK(hipEventRecord(a, 0));
            hipLaunchKernelGGL(k_read, dim3(8), dim3(256), 0, 0, (const uint32_t*)peer, dst, n);
            CHECK(hipEventRecord(b, 0));
            CHECK(hipDeviceSynchronize());
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, a, b));
            printf("[reader] kernel over the mapped buffer: %.1f us\n", ms * 1e3);
        }
        uint32_t* host = (uint32_t*)malloc(n * 4);
        CHECK(hipMemcpy(host, dst, n * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < n; ++i) bad += host[i] != (uint32_t)(i * 2654435761u) + 1u;
        printf("[reader] %zu of %zu words wrong\n", bad, n);
        CHECK(hipIpcCloseMemHandle(peer));
        fflush(stdout);
        char ok = bad == 0;
        (void)!write(to_parent[1], &ok, 1);
        _exit(bad ? 5 : 0);
    }
    const char* who = "owner";
    CHECK(hipSetDevice(0));
    uint32_t* buf;
    CHECK(hipMalloc(&buf, n * 4));
    uint32_t* host = (uint32_t*)malloc(n * 4);
    for (size_t i = 0; i < n; ++i) host[i] = (uint32_t)(i * 2654435761u);
    CHECK(hipMemcpy(buf, host, n * 4, hipMemcpyHostToDevice));
    hipIpcMemHandle_t h;
    CHECK(hipIpcGetMemHandle(&h, buf));
    printf("[owner] handle exported (%zu bytes)\n", sizeof(h));
    fflush(stdout);
    if (write(to_child[1], &h, sizeof(h)) != (ssize_t)sizeof(h)) return 2;
    char ok = 0;
    (void)!read(to_parent[0], &ok, 1);
    int st = 0;
    waitpid(pid, &st, 0);
    printf("[owner] reader exit %d, data %s\n", WEXITSTATUS(st), ok ? "identical" : "NOT verified");
    return ok ? 0 : 1;
}
