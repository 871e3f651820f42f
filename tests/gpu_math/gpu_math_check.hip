// gpu_math_check.hip — TEST INFRASTRUCTURE ONLY: runs device functions of rpf_math.h that have no bit-identical host
// twin (they use gfx950's approximate v_rsq_f32 / v_rcp_f32) over whole ranges of binary32 inputs ON THE GPU and counts
// the results that differ from the reference expression.  Built and loaded by tests/test_gpu_math.py.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../rawphotoforge_amd/csrc/rpf_fast.h"
#include "../../rawphotoforge_amd/csrc/rpf_math.h"

// sqrt_cr(s) against the compiler's IEEE expansion of sqrtf for the bit patterns [lo, lo + n)
__global__ void k_check_sqrt(uint32_t lo, uint64_t n, unsigned long long* bad, uint32_t* first_bad)
{
    unsigned long long mine = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const float s = rpf::bits2f(lo + (uint32_t)i);
        const float got = rpf::sqrt_cr(s);
        float ref;
        {
            volatile float sv = s;  // keep the reference on the plain sqrtf path
            ref = __builtin_sqrtf(sv);
        }
        const bool same = rpf::f2bits(got) == rpf::f2bits(ref) || (got != got && ref != ref);
        if (!same) {
            ++mine;
            atomicMin(first_bad, lo + (uint32_t)i);
        }
    }
    if (mine) atomicAdd(bad, mine);
}

// pow_1p5(x) against the binary64 formulation (float)(x * sqrt(x)) with the compiler's IEEE square root
__global__ void k_check_pow_1p5(uint32_t lo, uint64_t n, unsigned long long* bad, uint32_t* first_bad)
{
    unsigned long long mine = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const float x = rpf::bits2f(lo + (uint32_t)i);
        const float got = rpf::pow_1p5(x);
        volatile double xv = (double)x;
        const float ref = (float)(xv * __builtin_sqrt(xv));
        if (rpf::f2bits(got) != rpf::f2bits(ref)) {
            ++mine;
            atomicMin(first_bad, lo + (uint32_t)i);
        }
    }
    if (mine) atomicAdd(bad, mine);
}

template <class K>
static int run_check(K kernel, uint32_t lo_bits, uint64_t n, uint64_t* mismatches, uint32_t* first_bad_bits)
{
    unsigned long long* d_bad = nullptr;
    uint32_t* d_first = nullptr;
    if (hipMalloc((void**)&d_bad, sizeof(*d_bad)) != hipSuccess) return 1;
    if (hipMalloc((void**)&d_first, sizeof(*d_first)) != hipSuccess) return 1;
    const uint32_t ff = 0xffffffffu;
    (void)hipMemset(d_bad, 0, sizeof(*d_bad));
    (void)hipMemcpy(d_first, &ff, sizeof(ff), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(kernel, dim3(256 * 8), dim3(256), 0, 0, lo_bits, n, d_bad, d_first);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    unsigned long long bad = 0;
    (void)hipMemcpy(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost);
    (void)hipMemcpy(first_bad_bits, d_first, sizeof(uint32_t), hipMemcpyDeviceToHost);
    (void)hipFree(d_bad);
    (void)hipFree(d_first);
    *mismatches = bad;
    return 0;
}

extern "C" int gpu_check_pow_1p5(uint32_t lo_bits, uint64_t n, uint64_t* mismatches, uint32_t* first_bad_bits)
{
    return run_check(k_check_pow_1p5, lo_bits, n, mismatches, first_bad_bits);
}

extern "C" int gpu_check_sqrt(uint32_t lo_bits, uint64_t n, uint64_t* mismatches, uint32_t* first_bad_bits)
{
    unsigned long long* d_bad = nullptr;
    uint32_t* d_first = nullptr;
    if (hipMalloc((void**)&d_bad, sizeof(*d_bad)) != hipSuccess) return 1;
    if (hipMalloc((void**)&d_first, sizeof(*d_first)) != hipSuccess) return 1;
    const uint32_t ff = 0xffffffffu;
    (void)hipMemset(d_bad, 0, sizeof(*d_bad));
    (void)hipMemcpy(d_first, &ff, sizeof(ff), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_check_sqrt, dim3(256 * 8), dim3(256), 0, 0, lo_bits, n, d_bad, d_first);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    unsigned long long bad = 0;
    (void)hipMemcpy(&bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost);
    (void)hipMemcpy(first_bad_bits, d_first, sizeof(uint32_t), hipMemcpyDeviceToHost);
    (void)hipFree(d_bad);
    (void)hipFree(d_first);
    *mismatches = bad;
    return 0;
}

// hue_index on the device (v_rcp_f32 in the candidate, the sparse table lookup) for n directions; bounds = the 65536
// boundary directions (c, s) the host built (tests pass the twin's table).  neg_zero_x = 0 is the chain's call.
template <bool NEG_ZERO_X>
__global__ void k_hue_index(const float* y, const float* x, uint64_t n, const rpf::HueBound* hb, uint32_t* out)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = rpf::hue_index<NEG_ZERO_X>(y[i], x[i], hb);
}

extern "C" int gpu_hue_index(const float* y, const float* x, uint64_t n, const double* bounds, int neg_zero_x, uint32_t* out)
{
    float *dy = nullptr, *dx = nullptr;
    rpf::HueBound* dhb = nullptr;
    uint32_t* dout = nullptr;
    static_assert(sizeof(rpf::HueBound) == 16, "HueBound is two doubles");
    if (hipMalloc((void**)&dy, n * 4) != hipSuccess || hipMalloc((void**)&dx, n * 4) != hipSuccess ||
        hipMalloc((void**)&dout, n * 4) != hipSuccess || hipMalloc((void**)&dhb, 65536 * 16) != hipSuccess)
        return 1;
    (void)hipMemcpy(dy, y, n * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dx, x, n * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dhb, bounds, 65536 * 16, hipMemcpyHostToDevice);
    if (neg_zero_x) hipLaunchKernelGGL(k_hue_index<true>, dim3(2048), dim3(256), 0, 0, dy, dx, n, dhb, dout);
    else hipLaunchKernelGGL(k_hue_index<false>, dim3(2048), dim3(256), 0, 0, dy, dx, n, dhb, dout);
    const int rc = hipDeviceSynchronize() == hipSuccess ? 0 : 2;
    (void)hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost);
    (void)hipFree(dy); (void)hipFree(dx); (void)hipFree(dout); (void)hipFree(dhb);
    return rc;
}

// hue_candidate's error, exhaustively.  The candidate depends on the direction (x, y) only through q = min / max in
// [0, 1] (the polynomial) and through the octant (three exact reflections about 16383.75, 32767.5, 65535, each one more
// binary32 rounding of a value below 65536: <= 2^-8 / 2 each).  With y = q and x = 1 the device code's reciprocal is
// exact, no reflection is taken, and the sweep visits EVERY binary32 q in [0, 1]: max |candidate - atan(q) 65535 / (2 pi)|.
__global__ void k_hue_candidate_sweep(uint32_t lo, uint64_t n, unsigned int* max_err_bits)
{
    float worst = 0.0f;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const float q = rpf::bits2f(lo + (uint32_t)i);
        float mx;
        const float v = rpf::hue_candidate(q, 1.0f, mx);
        const double truth = atan((double)q) * (65535.0 / 6.283185307179586476925286766559);
        const float e = (float)fabs((double)v - truth);
        worst = e > worst ? e : worst;
    }
    atomicMax(max_err_bits, rpf::f2bits(worst));  // non-negative floats order like their bit patterns
}

extern "C" int gpu_hue_candidate_sweep(uint32_t lo_bits, uint64_t n, float* max_err)
{
    unsigned int* d = nullptr;
    if (hipMalloc((void**)&d, 4) != hipSuccess) return 1;
    (void)hipMemset(d, 0, 4);
    hipLaunchKernelGGL(k_hue_candidate_sweep, dim3(256 * 8), dim3(256), 0, 0, lo_bits, n, d);
    const int rc = hipDeviceSynchronize() == hipSuccess ? 0 : 2;
    unsigned int bits = 0;
    (void)hipMemcpy(&bits, d, 4, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    *max_err = rpf::bits2f(bits);
    return rc;
}


// ---- RPF_PRECISION_FAST builtins (rpf_fast.h) on the device ----------------------------------------------------------

// v_sqrt_f32 against the IEEE square root: the largest distance in ulp over the bit patterns [lo, lo + n) (positive normals)
__global__ void k_fast_sqrt_sweep(uint32_t lo, uint64_t n, unsigned int* max_ulp)
{
    unsigned int worst = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const float s = rpf::bits2f(lo + (uint32_t)i);
        const float got = rpf::sqrt_fast(s);
        volatile float sv = s;
        const float ref = __builtin_sqrtf(sv);
        const int d = (int)rpf::f2bits(got) - (int)rpf::f2bits(ref);  // both positive: patterns order like values
        const unsigned int a = (unsigned int)(d < 0 ? -d : d);
        worst = a > worst ? a : worst;
    }
    atomicMax(max_ulp, worst);
}

extern "C" int gpu_fast_sqrt_sweep(uint32_t lo_bits, uint64_t n, uint32_t* max_ulp)
{
    unsigned int* d = nullptr;
    if (hipMalloc((void**)&d, 4) != hipSuccess) return 1;
    (void)hipMemset(d, 0, 4);
    hipLaunchKernelGGL(k_fast_sqrt_sweep, dim3(256 * 8), dim3(256), 0, 0, lo_bits, n, d);
    const int rc = hipDeviceSynchronize() == hipSuccess ? 0 : 2;
    (void)hipMemcpy(max_ulp, d, 4, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    return rc;
}

// hue_position_fast on the device (v_rcp_f32 inside) for n directions
__global__ void k_fast_hue_position(const float* y, const float* x, uint64_t n, float* out)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = rpf::hue_position_fast(y[i], x[i]);
}

extern "C" int gpu_fast_hue_position(const float* y, const float* x, uint64_t n, float* out)
{
    float *dy = nullptr, *dx = nullptr, *dout = nullptr;
    if (hipMalloc((void**)&dy, n * 4) != hipSuccess || hipMalloc((void**)&dx, n * 4) != hipSuccess ||
        hipMalloc((void**)&dout, n * 4) != hipSuccess)
        return 1;
    (void)hipMemcpy(dy, y, n * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dx, x, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_fast_hue_position, dim3(2048), dim3(256), 0, 0, dy, dx, n, dout);
    const int rc = hipDeviceSynchronize() == hipSuccess ? 0 : 2;
    (void)hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost);
    (void)hipFree(dy); (void)hipFree(dx); (void)hipFree(dout);
    return rc;
}

__global__ void k_fast_hue_fraction(const float* y, const float* x, uint64_t n, float* out)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = rpf::hue_fraction_fast(y[i], x[i]);
}

extern "C" int gpu_fast_hue_fraction(const float* y, const float* x, uint64_t n, float* out)
{
    float *dy = nullptr, *dx = nullptr, *dout = nullptr;
    if (hipMalloc((void**)&dy, n * 4) != hipSuccess || hipMalloc((void**)&dx, n * 4) != hipSuccess ||
        hipMalloc((void**)&dout, n * 4) != hipSuccess)
        return 1;
    (void)hipMemcpy(dy, y, n * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dx, x, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_fast_hue_fraction, dim3(2048), dim3(256), 0, 0, dy, dx, n, dout);
    const int rc = hipDeviceSynchronize() == hipSuccess ? 0 : 2;
    (void)hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost);
    (void)hipFree(dy); (void)hipFree(dx); (void)hipFree(dout);
    return rc;
}

// powc_fast (which = 0: the cube root) / srgb_upper_fast (which = 1) over the bit patterns [lo, lo + n) with the table the host built, and
// sincos_hue_fast for all 65536 indices: IEEE operations only, so these must be the host twin's bits
__global__ void k_fast_pow_range(int which, uint32_t lo, uint64_t n, const rpf::FastTables* T, float* out)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const float x = rpf::bits2f(lo + (uint32_t)i);
        out[i] = which == 0 ? rpf::powc_fast<rpf::POW_CBRT>(x, &T->cbrt[0]) : rpf::srgb_upper_fast(x, &T->gamma[0]);
    }
}

extern "C" int gpu_fast_pow_range(int which, uint32_t lo_bits, uint64_t n, const void* host_tables, float* out)
{
    rpf::FastTables* dT = nullptr;
    float* dout = nullptr;
    if (hipMalloc((void**)&dT, sizeof(rpf::FastTables)) != hipSuccess || hipMalloc((void**)&dout, n * 4) != hipSuccess) return 1;
    (void)hipMemcpy(dT, host_tables, sizeof(rpf::FastTables), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_fast_pow_range, dim3(2048), dim3(256), 0, 0, which, lo_bits, n, dT, dout);
    const int rc = hipDeviceSynchronize() == hipSuccess ? 0 : 2;
    (void)hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost);
    (void)hipFree(dT); (void)hipFree(dout);
    return rc;
}

__global__ void k_fast_sincos_all(float* cs, float* sn)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    rpf::sincos_hue_fast((float)k, cs[k], sn[k]);
}

extern "C" int gpu_fast_sincos_all(float* cs, float* sn)
{
    float* d = nullptr;
    if (hipMalloc((void**)&d, 2 * 65536 * 4) != hipSuccess) return 1;
    hipLaunchKernelGGL(k_fast_sincos_all, dim3(256), dim3(256), 0, 0, d, d + 65536);
    const int rc = hipDeviceSynchronize() == hipSuccess ? 0 : 2;
    (void)hipMemcpy(cs, d, 65536 * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(sn, d + 65536, 65536 * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    return rc;
}
