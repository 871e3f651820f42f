// gather_probe.hip — what does a 64-lane table gather cost on gfx950?  (measurement tool, not product)
//   hipcc --offload-arch=gfx950 -O3 tools/gather_probe.hip -o tools/gather_probe && tools/gather_probe
// Every wave issues ITERS rounds of 4 independent gathers of ESZ bytes per lane and sums what it reads; 20 waves per CU
// (5 workgroups of 256 threads, the fused kernel's residency), all CUs.  Index patterns per gather:
//   random    every lane its own pseudo-random entry of the table (64 distinct lines when the table is large)
//   sparse    3 lanes of 8 random, the others entry 0 (the sparse hue lookup)
//   oneline   all lanes inside ONE 128-byte line (different entries of it), a different line per gather
//   same      all lanes the same entry
//   stride    lane i reads entry base + i (contiguous: 64 * ESZ bytes)
// Reported: nanoseconds per gather instruction per CU, and distinct 128-byte lines per nanosecond per CU.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define ITERS 400

template <int ESZ> struct Elem;
template <> struct Elem<2> { typedef uint16_t T; };
template <> struct Elem<8> { typedef uint2 T; };
template <> struct Elem<16> { typedef uint4 T; };

__device__ __forceinline__ uint32_t fold(uint16_t v) { return v; }
__device__ __forceinline__ uint32_t fold(uint2 v) { return v.x ^ v.y; }
__device__ __forceinline__ uint32_t fold(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

template <int ESZ, int PATTERN>
__global__ __launch_bounds__(256) void k_gather(const unsigned char* table, uint32_t nentries, uint32_t* out)
{
    typedef typename Elem<ESZ>::T T;
    const T* tab = reinterpret_cast<const T*>(table);
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t rnd = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    uint32_t wrnd = (blockIdx.x * 4u + (threadIdx.x >> 6)) * 2246822519u + 777u;  // wave-uniform stream
    uint32_t acc = 0;
    constexpr uint32_t PER_LINE = 128u / ESZ;
    for (int it = 0; it < ITERS; ++it) {
        uint32_t idx[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            rnd = rnd * 1664525u + 1013904223u;
            wrnd = wrnd * 1664525u + 1013904223u;
            const uint32_t r = (rnd >> 8) % nentries;
            const uint32_t w = (wrnd >> 8) % nentries;
            if (PATTERN == 0) idx[u] = r;
            if (PATTERN == 1) idx[u] = ((rnd >> 29) < 3u) ? r : 0u;                         // 3/8 of the lanes
            if (PATTERN == 2) idx[u] = (w / PER_LINE) * PER_LINE + (lane % PER_LINE);         // one line
            if (PATTERN == 3) idx[u] = w;                                                     // one entry
            if (PATTERN == 4) idx[u] = (w / 64u) * 64u % (nentries - 64u) + lane;             // contiguous
        }
        T v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = tab[idx[u]];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += fold(v[u]);
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}

// The random 16-byte gather again through a buffer descriptor, with the load's cache-policy bits (gfx940+: aux bit 0 =
// sc0, bit 1 = nt, bit 4 = sc1) as a template parameter: does any policy make a missing line cheaper than a 128-byte fill?
template <int AUX>
__global__ __launch_bounds__(256) void k_gather_policy(const unsigned char* table, uint32_t nentries, uint32_t* out)
{
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(table), 0, (int)(nentries * 16u), 0x00020000);
    uint32_t rnd = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    for (int it = 0; it < ITERS; ++it) {
        uint32_t idx[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            rnd = rnd * 1664525u + 1013904223u;
            idx[u] = (rnd >> 8) % nentries;
        }
        v4u v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, idx[u] * 16u, 0, AUX);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}

template <int AUX>
static void run_policy(const char* what)
{
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, blocks = cus * 5;
    const size_t table_bytes = 2048 * 1024;
    unsigned char* tab; uint32_t* out;
    (void)hipMalloc(&tab, table_bytes);
    (void)hipMemset(tab, 1, table_bytes);
    (void)hipMalloc(&out, sizeof(uint32_t) * blocks * 256);
    const uint32_t n = (uint32_t)(table_bytes / 16);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k_gather_policy<AUX>), dim3(blocks), dim3(256), 0, 0, tab, n, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_gather_policy<AUX>), dim3(blocks), dim3(256), 0, 0, tab, n, out);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / (20.0 * ITERS * 4.0);
    printf("random 16 B/lane, 2048 KiB table, buffer load, policy %-12s: %7.2f ns per gather per CU = %5.2f distinct lines / ns / CU\n",
           what, ns, 62.0 / ns);
    (void)hipFree(tab); (void)hipFree(out);
}

template <int ESZ, int PATTERN>
static void run(const char* what, size_t table_bytes, double lines_per_gather)
{
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, blocks = cus * 5;
    unsigned char* tab; uint32_t* out;
    (void)hipMalloc(&tab, table_bytes);
    (void)hipMemset(tab, 1, table_bytes);
    (void)hipMalloc(&out, sizeof(uint32_t) * blocks * 256);
    const uint32_t n = (uint32_t)(table_bytes / ESZ);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k_gather<ESZ, PATTERN>), dim3(blocks), dim3(256), 0, 0, tab, n, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_gather<ESZ, PATTERN>), dim3(blocks), dim3(256), 0, 0, tab, n, out);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double gathers_per_cu = 20.0 * ITERS * 4.0;   // wave-instructions per CU
    const double ns = ms * 1e6 / gathers_per_cu;
    printf("%-8s %2d B/lane  table %7.0f KiB: %7.2f ns per gather per CU", what, ESZ, table_bytes / 1024.0, ns);
    if (lines_per_gather > 0) printf("  = %5.2f distinct lines / ns / CU", lines_per_gather / ns);
    printf("\n");
    (void)hipFree(tab); (void)hipFree(out);
}

int main(int argc, char** argv)
{
    if (argc > 1 && argv[1][0] == 'p') {   // gather_probe policy
        run_policy<0>("default");
        run_policy<1>("sc0");
        run_policy<2>("nt");
        run_policy<3>("sc0 nt");
        run_policy<16>("sc1");
        run_policy<17>("sc0 sc1");
        run_policy<18>("sc1 nt");
        run_policy<19>("sc0 sc1 nt");
        return 0;
    }
    for (size_t kb : {16, 128, 1024, 2048}) {
        run<16, 0>("random", kb * 1024, kb >= 128 ? 62.0 : 0);
        run<8, 0>("random", kb * 1024, kb >= 128 ? 62.0 : 0);
        run<2, 0>("random", kb * 1024, kb >= 128 ? 62.0 : 0);
    }
    run<16, 1>("sparse", 1024 * 1024, 25.0);
    run<16, 2>("oneline", 1024 * 1024, 1.0);
    run<2, 2>("oneline", 128 * 1024, 1.0);
    run<16, 3>("same", 1024 * 1024, 1.0);
    run<2, 3>("same", 128 * 1024, 1.0);
    run<16, 4>("stride", 1024 * 1024, 8.0);
    run<2, 4>("stride", 128 * 1024, 1.0);
    return 0;
}
