// mfma_probe.hip — can the matrix pipe take the colour-matrix multiplies off the VALU?  (measurement tool, not product)
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
// v_mfma_f32_4x4x1_16b_f32 with A = (c0, c1, c2, c3) repeated over the lanes (lane % 4 selects) and B = one value per
// lane returns, in its four destination registers, c_i * b(lane) + C_i for i = 0..3 — four products of one per-lane value
// per instruction, on the matrix pipe.  Part 1: is RN(a*b + (-0)) from that pipe bit-identical to v_mul_f32 (denormals,
// signed zeros, infinities, NaN payloads aside)?  Part 2: what does an MFMA cost next to VALU work of the same waves:
// loop bodies of 16 v_mul_f32, the same plus 2 / 4 / 8 MFMAs, and MFMAs alone, at 4 and 8 waves per SIMD.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- part 1: numerics
__global__ void k_numerics(const float* consts, const float* xs, float* mul_out, float* mfma_out, float* mfma0_out,
                           float* chain_out, int n)
{
    const int lane = threadIdx.x & 63;
    const float a = consts[lane & 3];
    const v4f negz = {-0.0f, -0.0f, -0.0f, -0.0f};
    const v4f zero = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float b = xs[i];
        const v4f d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, negz, 0, 0, 0);
        const v4f d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, zero, 0, 0, 0);
        // chain: is a*b + c one rounding (fma) or two?
        const v4f cc = {b, b, b, b};
        const v4f dc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, cc, 0, 0, 0);
        for (int k = 0; k < 4; ++k) {
            float m;
            const float ck = consts[k];
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m) : "v"(ck), "v"(b));
            mul_out[(size_t)i * 4 + k] = m;
            mfma_out[(size_t)i * 4 + k] = d[k];
            mfma0_out[(size_t)i * 4 + k] = d0[k];
            chain_out[(size_t)i * 4 + k] = dc[k];
        }
    }
}

// ---------------------------------------------------------------- part 2: issue cost
#define ITERS 2000
#define MUL4(a, b, c, d) "v_mul_f32 %" #a ", %12, %" #a "\nv_mul_f32 %" #b ", %12, %" #b "\nv_mul_f32 %" #c ", %12, %" #c "\nv_mul_f32 %" #d ", %12, %" #d "\n"
#define MF(d) "v_mfma_f32_4x4x1_16b_f32 %" #d ", %12, %13, %14\n"
#define OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3) \
            : "v"(x), "v"(y), "v"(c)

template <int NMUL16, int NMFMA>
__global__ void k_mix(float* out, long long* cyc, float x, float y)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v4f m0 = {0, 0, 0, 0}, m1 = m0, m2 = m0, m3 = m0;
    const v4f c = {-0.0f, -0.0f, -0.0f, -0.0f};
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITERS; ++i) {
        if (NMUL16 == 1 && NMFMA == 0)
            asm volatile(MUL4(0, 1, 2, 3) MUL4(4, 5, 6, 7) MUL4(0, 1, 2, 3) MUL4(4, 5, 6, 7)
                         OPS);
        if (NMUL16 == 1 && NMFMA == 2)
            asm volatile(MUL4(0, 1, 2, 3) MUL4(4, 5, 6, 7) MF(8) MUL4(0, 1, 2, 3) MUL4(4, 5, 6, 7) MF(9)
                         OPS);
        if (NMUL16 == 1 && NMFMA == 4)
            asm volatile(MUL4(0, 1, 2, 3) MF(8) MUL4(4, 5, 6, 7) MF(9) MUL4(0, 1, 2, 3) MF(10) MUL4(4, 5, 6, 7) MF(11)
                         OPS);
        if (NMUL16 == 1 && NMFMA == 8)
            asm volatile(MUL4(0, 1, 2, 3) MF(8) MF(9) MUL4(4, 5, 6, 7) MF(10) MF(11) MUL4(0, 1, 2, 3) MF(8) MF(9)
                             MUL4(4, 5, 6, 7) MF(10) MF(11)
                         OPS);
        if (NMUL16 == 0 && NMFMA == 4)
            asm volatile(MF(8) MF(9) MF(10) MF(11)
                         OPS);
        if (NMUL16 == 0 && NMFMA == 8)
            asm volatile(MF(8) MF(9) MF(10) MF(11) MF(8) MF(9) MF(10) MF(11)
                         OPS);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + m0[0] + m1[1] + m2[2] + m3[3];
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <class K>
static void run(const char* name, K kernel)
{
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("%-22s", name);
    for (int wps : {1, 4, 8}) {
        const size_t lds = (size_t)(160 * 1024) / wps - 64;
        const int blocks = cus * wps;
        float* out; long long* cyc;
        (void)hipMalloc(&out, sizeof(float) * blocks * 256);
        (void)hipMalloc(&cyc, sizeof(long long) * blocks * 4);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), lds, 0, out, cyc, 1.0001f, 0.5f);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), lds, 0, out, cyc, 1.0001f, 0.5f);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("  %d w/SIMD: %7.2f ns/body", wps, ms * 1e6 / ITERS / wps);   // per wave-body per SIMD
        (void)hipFree(out); (void)hipFree(cyc);
    }
    printf("\n");
}

int main()
{
    // ---- numerics
    const int n = 1 << 22;
    std::vector<float> xs(n);
    uint64_t s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        uint32_t b = (uint32_t)(s >> 20);
        if (i % 8 == 1) b &= 0x807fffffu;                       // denormal inputs
        if (i % 8 == 2) b = (b & 0x80ffffffu) | 0x00800000u;    // tiny normals: products underflow
        if (i % 8 == 3) b = (b & 0x807fffffu) | 0x7f000000u;    // huge: products overflow
        if (i % 64 == 5) b = (b & 0x80000000u);                 // signed zeros
        memcpy(&xs[i], &b, 4);
    }
    const float consts[4] = {0.4122214708f, -1.2684380046f, 3.0e-39f /* denormal constant */, -4.0767416621f};
    float *dc, *dx, *o0, *o1, *o2, *o3;
    (void)hipMalloc(&dc, 16); (void)hipMalloc(&dx, 4ull * n);
    (void)hipMalloc(&o0, 16ull * n); (void)hipMalloc(&o1, 16ull * n); (void)hipMalloc(&o2, 16ull * n); (void)hipMalloc(&o3, 16ull * n);
    (void)hipMemcpy(dc, consts, 16, hipMemcpyHostToDevice);
    (void)hipMemcpy(dx, xs.data(), 4ull * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_numerics, dim3(1024), dim3(256), 0, 0, dc, dx, o0, o1, o2, o3, n);
    (void)hipDeviceSynchronize();
    std::vector<uint32_t> h0(4ull * n), h1(4ull * n), h2(4ull * n), h3(4ull * n);
    (void)hipMemcpy(h0.data(), o0, 16ull * n, hipMemcpyDeviceToHost);
    (void)hipMemcpy(h1.data(), o1, 16ull * n, hipMemcpyDeviceToHost);
    (void)hipMemcpy(h2.data(), o2, 16ull * n, hipMemcpyDeviceToHost);
    (void)hipMemcpy(h3.data(), o3, 16ull * n, hipMemcpyDeviceToHost);
    long bad = 0, bad_nonan = 0, bad0 = 0, bad_den = 0, fused = 0, unfused = 0, neither = 0;
    for (size_t i = 0; i < 4ull * n; ++i) {
        float m, f; memcpy(&m, &h0[i], 4); memcpy(&f, &h1[i], 4);
        const bool nan = m != m;
        if (h0[i] != h1[i]) {
            ++bad;
            if (!nan || f == f) {
                ++bad_nonan;
                if ((h0[i] & 0x7f800000u) == 0 || (h1[i] & 0x7f800000u) == 0) ++bad_den;
                if (bad_nonan <= 8) {
                    float x = xs[i / 4];
                    uint32_t xb; memcpy(&xb, &x, 4);
                    printf("  mismatch: c[%zu] * %08x -> v_mul %08x  mfma %08x\n", i % 4, xb, h0[i], h1[i]);
                }
            }
        }
        if (h0[i] != h2[i] && !nan) ++bad0;
        // chain: c*b + b
        const float c = consts[i % 4], b = xs[i / 4];
        const float fu = fmaf(c, b, b);
        volatile float p = c * b;
        const float un = p + b;
        uint32_t fub, unb; memcpy(&fub, &fu, 4); memcpy(&unb, &un, 4);
        if (fu != fu) continue;
        if (h3[i] == fub && h3[i] != unb) ++fused;
        else if (h3[i] == unb && h3[i] != fub) ++unfused;
        else if (h3[i] != unb && h3[i] != fub) ++neither;
    }
    printf("numerics over %d values x 4 constants: mfma(C=-0) != v_mul: %ld (non-NaN: %ld, of them with a denormal result: %ld); "
           "mfma(C=+0) != v_mul (non-NaN): %ld\n", n, bad, bad_nonan, bad_den, bad0);
    printf("a*b + c in the matrix pipe: matches fma only %ld, mul-then-add only %ld, neither %ld\n", fused, unfused, neither);

    // ---- issue cost
    run("16 mul", k_mix<1, 0>);
    run("16 mul + 2 mfma", k_mix<1, 2>);
    run("16 mul + 4 mfma", k_mix<1, 4>);
    run("16 mul + 8 mfma", k_mix<1, 8>);
    run("4 mfma", k_mix<0, 4>);
    run("8 mfma", k_mix<0, 8>);
    return 0;
}
