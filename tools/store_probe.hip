// store_probe.hip — does the order in which a frame's 1-KiB row pieces are written matter?  (measurement tool)
//   hipcc --offload-arch=gfx950 -O3 tools/store_probe.hip -o tools/store_probe && tools/store_probe
// 384 MB (6000x4000 RGBA32F: 4000 rows of 96 000 B) written with nontemporal 16-byte stores, 1 KiB per wave-instruction:
//   linear   wave-instruction k of the launch writes KiB k (a fill)
//   strips   a wave owns a 64-pixel strip (1 KiB per row) and walks down 32 rows (the fused kernel's item), items dealt
//            round-robin to the resident waves
//   strips2  the same, but a wave owns TWO neighbouring strips and writes both rows pieces back to back (2 KiB contiguous)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

typedef uint32_t v4u __attribute__((ext_vector_type(4)));
constexpr int W = 6000, H = 4000;
constexpr size_t ROWB = (size_t)W * 16;

__global__ __launch_bounds__(256) void k_linear(unsigned char* out, size_t nkib)
{
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (size_t)gridDim.x * 4;
    const uint32_t lane = threadIdx.x & 63;
    for (size_t k = wave; k < nkib; k += nw)
        __builtin_nontemporal_store(v4u{1, 2, 3, 4}, reinterpret_cast<v4u*>(out + k * 1024 + lane * 16));
}

template <int NSTRIP>
__global__ __launch_bounds__(256) void k_strips(unsigned char* out)
{
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4, lane = threadIdx.x & 63;
    const uint32_t strips = (W + 63) / 64 / NSTRIP, segs = H / 32;
    for (uint32_t item = wave; item < strips * segs; item += nw) {
        const uint32_t seg = item / strips, strip = item - seg * strips;
        for (uint32_t r = 0; r < 32; ++r)
            for (int s = 0; s < NSTRIP; ++s) {
                const size_t col = ((size_t)strip * NSTRIP + s) * 64 + lane;
                if (col < (size_t)W)
                    __builtin_nontemporal_store(v4u{1, 2, 3, 4}, reinterpret_cast<v4u*>(out + (size_t)(seg * 32 + r) * ROWB + col * 16));
            }
    }
}

// pairs: a wave owns a 128-pixel strip (2 KiB per row); lane l writes pixels 2l and 2l + 1 with TWO instructions, each of
// which covers every other 16 bytes of the 2 KiB (what a lane that owns two adjacent Bayer columns would store)
template <bool NT>
__global__ __launch_bounds__(256) void k_pairs(unsigned char* out)
{
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4, lane = threadIdx.x & 63;
    const uint32_t strips = (W + 127) / 128, segs = H / 32;
    for (uint32_t item = wave; item < strips * segs; item += nw) {
        const uint32_t seg = item / strips, strip = item - seg * strips;
        for (uint32_t r = 0; r < 32; ++r)
            for (int s = 0; s < 2; ++s) {
                const size_t col = (size_t)strip * 128 + lane * 2 + s;
                if (col < (size_t)W) {
                    v4u* p = reinterpret_cast<v4u*>(out + (size_t)(seg * 32 + r) * ROWB + col * 16);
                    if (NT) __builtin_nontemporal_store(v4u{1, 2, 3, 4}, p);
                    else *p = v4u{1, 2, 3, 4};
                }
            }
    }
}
// the pairs pattern through raw buffer stores with each cache-policy combination (aux: 1 = sc0, 2 = nt, 16 = sc1)
template <int AUX>
__global__ __launch_bounds__(256) void k_pairs_aux(unsigned char* out)
{
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4, lane = threadIdx.x & 63;
    const uint32_t strips = (W + 127) / 128, segs = H / 32;
    for (uint32_t item = wave; item < strips * segs; item += nw) {
        const uint32_t seg = item / strips, strip = item - seg * strips;
        for (uint32_t r = 0; r < 32; ++r) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)(seg * 32 + r) * ROWB, 0, (int)ROWB, 0x00020000);
            for (int s = 0; s < 2; ++s)
                __builtin_amdgcn_raw_buffer_store_b128(v4u{1, 2, 3, 4}, rs, (strip * 128 + lane * 2 + s) * 16, 0, AUX);
        }
    }
}
// swapped: what v_permlane32_swap leaves: ONE instruction covers a whole KiB, but lanes 0..31 hold its even pixels and
// lanes 32..63 its odd ones (two instructions per 2-KiB row piece)
template <int AUX>
__global__ __launch_bounds__(256) void k_swapped(unsigned char* out)
{
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4, lane = threadIdx.x & 63;
    const uint32_t strips = (W + 127) / 128, segs = H / 32;
    const uint32_t px = lane < 32 ? 2 * lane : 2 * (lane - 32) + 1;
    for (uint32_t item = wave; item < strips * segs; item += nw) {
        const uint32_t seg = item / strips, strip = item - seg * strips;
        for (uint32_t r = 0; r < 32; ++r) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)(seg * 32 + r) * ROWB, 0, (int)ROWB, 0x00020000);
            for (int s = 0; s < 2; ++s)
                __builtin_amdgcn_raw_buffer_store_b128(v4u{1, 2, 3, 4}, rs, (strip * 128 + s * 64 + px) * 16, 0, AUX);
        }
    }
}
// halves: the same 2 KiB per row, but instruction 1 writes its first KiB and instruction 2 its second (whole lines each)
__global__ __launch_bounds__(256) void k_halves(unsigned char* out)
{
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4, lane = threadIdx.x & 63;
    const uint32_t strips = (W + 127) / 128, segs = H / 32;
    for (uint32_t item = wave; item < strips * segs; item += nw) {
        const uint32_t seg = item / strips, strip = item - seg * strips;
        for (uint32_t r = 0; r < 32; ++r)
            for (int s = 0; s < 2; ++s) {
                const size_t col = (size_t)strip * 128 + s * 64 + lane;
                if (col < (size_t)W)
                    __builtin_nontemporal_store(v4u{1, 2, 3, 4}, reinterpret_cast<v4u*>(out + (size_t)(seg * 32 + r) * ROWB + col * 16));
            }
    }
}

template <class F>
static void timeit(const char* name, F launch)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) launch();
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < 50; ++i) launch();
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 50;
    printf("%-8s %.4f ms  %.2f TB/s\n", name, ms, (double)ROWB * H / ms / 1e9);
}

int main()
{
    unsigned char* buf;
    (void)hipMalloc(&buf, ROWB * H);
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int grid = p.multiProcessorCount * 5;
    timeit("linear", [&] { hipLaunchKernelGGL(k_linear, dim3(grid), dim3(256), 0, 0, buf, ROWB * H / 1024); });
    timeit("strips", [&] { hipLaunchKernelGGL(k_strips<1>, dim3(grid), dim3(256), 0, 0, buf); });
    timeit("strips2", [&] { hipLaunchKernelGGL(k_strips<2>, dim3(grid), dim3(256), 0, 0, buf); });
    timeit("strips4", [&] { hipLaunchKernelGGL(k_strips<4>, dim3(grid), dim3(256), 0, 0, buf); });
    timeit("halves", [&] { hipLaunchKernelGGL(k_halves, dim3(grid), dim3(256), 0, 0, buf); });
    timeit("pairs nt", [&] { hipLaunchKernelGGL(k_pairs<true>, dim3(grid), dim3(256), 0, 0, buf); });
    timeit("pairs", [&] { hipLaunchKernelGGL(k_pairs<false>, dim3(grid), dim3(256), 0, 0, buf); });
    timeit("swapped nt", [&] { hipLaunchKernelGGL(k_swapped<2>, dim3(grid), dim3(256), 0, 0, buf); });
    timeit("swapped", [&] { hipLaunchKernelGGL(k_swapped<0>, dim3(grid), dim3(256), 0, 0, buf); });
    timeit("pairs aux0", [&] { hipLaunchKernelGGL(k_pairs_aux<0>, dim3(grid), dim3(256), 0, 0, buf); });
    timeit("pairs sc0", [&] { hipLaunchKernelGGL(k_pairs_aux<1>, dim3(grid), dim3(256), 0, 0, buf); });
    timeit("pairs nt", [&] { hipLaunchKernelGGL(k_pairs_aux<2>, dim3(grid), dim3(256), 0, 0, buf); });
    timeit("pairs sc0nt", [&] { hipLaunchKernelGGL(k_pairs_aux<3>, dim3(grid), dim3(256), 0, 0, buf); });
    timeit("pairs sc1", [&] { hipLaunchKernelGGL(k_pairs_aux<16>, dim3(grid), dim3(256), 0, 0, buf); });
    timeit("pairs sc0sc1", [&] { hipLaunchKernelGGL(k_pairs_aux<17>, dim3(grid), dim3(256), 0, 0, buf); });
    timeit("pairs ntsc1", [&] { hipLaunchKernelGGL(k_pairs_aux<18>, dim3(grid), dim3(256), 0, 0, buf); });
    timeit("pairs all3", [&] { hipLaunchKernelGGL(k_pairs_aux<19>, dim3(grid), dim3(256), 0, 0, buf); });
    return 0;
}
