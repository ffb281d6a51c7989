// Device-side arithmetic shared by the HIP kernels.  Evaluation order and the
// fp32/fp64 mix follow the reference; everything is compiled with
// -ffp-contract=off.
#pragma once

#include <hip/hip_runtime.h>

#include "internal.h"

namespace fdtd {

typedef float f4 __attribute__((ext_vector_type(4)));

// E^{n+1} of one cell, fdtd_calculations.cu:150-156:
//   c * ( d*E^n - a*E^{n-1}                                   [fp32]
//         + 2(dt+tau)*D^{n+1} - (2dt+4tau)*D^n + 2tau*D^{n-1} [fp64] )
// The three fp64 factors are rebuilt from dt and tau exactly as the reference
// builds them (dt + tau in fp32, the rest in fp64; doubling is exact).
// 2(dt+tau) and 2tau are fp32 values held in fp64, so their products with an fp32 D (24 x 24 bits)
// are exact in fp64, and "acc + k*D" rounds once either way: those two multiply-add pairs are issued
// as one fused operation each, with the same bits as the reference's two.  (2dt + 4tau)*D^n is not
// exact in general and keeps its own rounding.
__device__ __forceinline__ float e_update(const MatEntry &m, float dt, float e0, float e1, float d0,
                                          float d1, float d2)
{
    const float f = m.d * e0 - m.a * e1;
    const double k1 = 2.0 * (double)(dt + m.tau);
    const double k2 = 2.0 * (double)dt + 4.0 * (double)m.tau;
    const double k3 = 2.0 * (double)m.tau;
    double acc = __builtin_fma(k1, (double)d0, (double)f);
    acc = acc - k2 * (double)d1;
    acc = __builtin_fma(k3, (double)d2, acc);
    return (float)((double)m.c * acc);
}

// The same with the three fp64 factors taken from the table (host_setup.cpp builds them with the
// operations above).
__device__ __forceinline__ float e_update_wide(const MatCoef &m, float e0, float e1, float d0, float d1, float d2)
{
    const float f = m.d * e0 - m.a * e1;
    double acc = __builtin_fma(m.k1, (double)d0, (double)f);
    acc = acc - m.k2 * (double)d1;
    acc = __builtin_fma(m.k3, (double)d2, acc);
    return (float)((double)m.c * acc);
}

// A grid of a single material: the factors of the E update are the same for every cell.  Held in
// scalar registers (fp64 arithmetic runs on the vector unit only, so the compiler, left alone,
// keeps such loop invariants in eight vector registers of every lane).
struct OneCoef {
    float a, d;
    double k1, k2, k3, c;
};

__device__ __forceinline__ double wave_uniform(double v)
{
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ OneCoef one_coef(const MatEntry &m, float dt)
{
    OneCoef o;
    o.a = m.a, o.d = m.d;
    o.k1 = wave_uniform(2.0 * (double)(dt + m.tau));
    o.k2 = wave_uniform(2.0 * (double)dt + 4.0 * (double)m.tau);
    o.k3 = wave_uniform(2.0 * (double)m.tau);
    o.c = wave_uniform((double)m.c);
    return o;
}

__device__ __forceinline__ float e_update_one(const OneCoef &m, float e0, float e1, float d0, float d1, float d2)
{
    const float f = m.d * e0 - m.a * e1;
    double acc = __builtin_fma(m.k1, (double)d0, (double)f);
    acc = acc - m.k2 * (double)d1;
    acc = __builtin_fma(m.k3, (double)d2, acc);
    return (float)(m.c * acc);
}

// First-order Mur face value, fdtd_calculations.cu:240-248 and its siblings.
// r_here: rp on the face layer, r_in: rp on the first interior layer.  The four factors that do not
// depend on the fields -- 1/(dt + s_here), dt - s_in, dt + s_in, dt - s_here -- can be taken once
// (mur_coefs) and applied every step (mur_apply): the same operations in the same order.
__device__ __forceinline__ f4 mur_coefs(float dt, float du, float mu0, float eps0, float r_here, float r_in)
{
    const float s_here = du * sqrtf(mu0 * eps0 * r_here);
    const float s_in = du * sqrtf(mu0 * eps0 * r_in);
    const f4 c = {1.0f / (dt + s_here), dt - s_in, dt + s_in, dt - s_here};
    return c;
}

__device__ __forceinline__ float mur_apply(const f4 &c, float et_in, float es_in, float es_here)
{
    return c.x * (c.y * et_in + c.z * es_in - c.w * es_here);
}

__device__ __forceinline__ float mur_face(float dt, float du, float mu0, float eps0, float r_here,
                                          float r_in, float et_in, float es_in, float es_here)
{
    return mur_apply(mur_coefs(dt, du, mu0, eps0, r_here, r_in), et_in, es_in, es_here);
}

// The field arrays are streamed once per step and are far larger than the
// caches: their loads and stores carry the nontemporal hint.
__device__ __forceinline__ f4 ld4_stream(const float *p) { return __builtin_nontemporal_load((const f4 *)p); }
__device__ __forceinline__ void st4_stream(float *p, f4 v) { __builtin_nontemporal_store(v, (f4 *)p); }

} // namespace fdtd
