// TEST INFRASTRUCTURE -- never linked into, loaded by or shipped with the product.
//
// Host driver around the REFERENCE's own kernel file.  oracle/Makefile (target `ref`, this
// container only) passes /root/reference/fdtd_calculations.cu and the three headers it includes
// through the image's hipify-perl into a scratch directory (cuda* runtime names become hip*;
// nothing else changes), puts that directory on the include path and compiles this file for
// gfx950 with -ffp-contract=off into oracle/_ref/libfdtd_ref.so.  The reference's sources are
// read where they lie; neither they nor their translated copies enter the repository.
//
// What runs on the GPU is therefore the reference's updateHField / updateDField /
// updateSources / updateEField / updateMurBoundary (fdtd_calculations.cu:32-428) and its
// copySymbolsToDevice (fdtd_calculations.cu:7-29), launched with main's geometry
// (fdtd.cu:13-15, 75-78) in main's order and buffer rotation (fdtd.cu:129-214 and its two
// unrolled repeats): step `it` writes buffer t = it % 3, reads s = (t + 2) % 3 as time n and
// o = (t + 1) % 3 as time n-1, and passes it - it % 3 to updateSources.
// tests/test_reference_kernels_gpu.py holds the CPU oracle (and through it the HIP path)
// against these kernels bit for bit.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "fdtd_calculations.cu" // the translated copy in the scratch directory

namespace {

struct DeviceArrays {
    std::vector<void *> all;
    ~DeviceArrays()
    {
        for (void *p : all) (void)hipFree(p);
    }
    // `slack` floats beyond the array: the reference's threads with ix >= nx read the material
    // arrays a few elements past the last row (fdtd_calculations.cu:141-144)
    float *up(const float *host, size_t n, bool *ok)
    {
        const size_t slack = 1024;
        float *d = nullptr;
        if (hipMalloc((void **)&d, (n + slack) * sizeof(float)) != hipSuccess) {
            *ok = false;
            return nullptr;
        }
        all.push_back(d);
        if (hipMemset(d, 0, (n + slack) * sizeof(float)) != hipSuccess ||
            hipMemcpy(d, host, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
            *ok = false;
        return d;
    }
};

} // namespace

// dims = nx, ny, nz; consts = dt, dx, dy, dz, mu0, eps0.
// h: 3 arrays of n = nx*ny*nz floats (hx, hy, hz).  e, d: [buffer 0..2][component x,y,z] = 9 arrays
// of n floats each, buffer-major.  mat: sigma, epsI, epsS, tauD, n floats each.  rp: the six planes
// in the order x0, y0, z0, xEnd, yEnd, zEnd with rp_count[f] floats.  Advances steps
// first_iteration .. first_iteration + nsteps - 1 and writes h, e, d back.
extern "C" int fdtd_ref_run(const int *dims, const float *consts, int nsrc, const int *sources, int jz_count,
                            const float *jz, float *h, float *e, float *d, const float *mat, const float *const *rp,
                            const size_t *rp_count, int first_iteration, int nsteps)
{
    const int nx = dims[0], ny = dims[1], nz = dims[2];
    const size_t n = (size_t)nx * ny * nz;
    FdtdParams p;
    std::memset(&p, 0, sizeof p);
    p.nx = nx;
    p.ny = ny;
    p.nz = nz;
    p.dt = consts[0];
    p.dx = consts[1];
    p.dy = consts[2];
    p.dz = consts[3];
    p.mu0 = consts[4];
    p.eps0 = consts[5];
    p.sourcesCount = nsrc;
    p.sources = const_cast<int *>(sources);
    p.jzCount = jz_count;
    p.jz = const_cast<float *>(jz);
    copySymbolsToDevice(&p);

    DeviceArrays dev;
    bool ok = true;
    float *dh[3], *de[3][3], *dd[3][3], *dm[4], *drp[6];
    for (int k = 0; k < 3; ++k) dh[k] = dev.up(h + k * n, n, &ok);
    for (int b = 0; b < 3; ++b)
        for (int k = 0; k < 3; ++k) {
            de[b][k] = dev.up(e + ((size_t)b * 3 + k) * n, n, &ok);
            dd[b][k] = dev.up(d + ((size_t)b * 3 + k) * n, n, &ok);
        }
    for (int k = 0; k < 4; ++k) dm[k] = dev.up(mat + k * n, n, &ok);
    for (int f = 0; f < 6; ++f) drp[f] = dev.up(rp[f], rp_count[f], &ok);
    if (!ok) return -1;

    // fdtd.cu:13-15, 75-78
    const dim3 block(128, 1, 1);
    const dim3 grid((nx + 128 - 1) / 128, ny, nz);
    for (int it = first_iteration; it < first_iteration + nsteps; ++it) {
        const int t = it % 3, s = (t + 2) % 3, o = (t + 1) % 3;
        hipLaunchKernelGGL(updateHField, grid, block, 0, 0, dh[0], dh[1], dh[2], de[s][0], de[s][1], de[s][2]);
        hipLaunchKernelGGL(updateDField, grid, block, 0, 0, dd[t][0], dd[t][1], dd[t][2], dd[s][0], dd[s][1], dd[s][2],
                           dh[0], dh[1], dh[2]);
        hipLaunchKernelGGL(updateSources, grid, block, 0, 0, dd[t][2], dd[s][2], dh[0], dh[1], it - it % 3);
        // sigma, epsI, epsS, tauD
        hipLaunchKernelGGL(updateEField, grid, block, 0, 0, de[t][0], de[t][1], de[t][2], de[s][0], de[s][1], de[s][2],
                           de[o][0], de[o][1], de[o][2], dd[t][0], dd[t][1], dd[t][2], dd[s][0], dd[s][1], dd[s][2],
                           dd[o][0], dd[o][1], dd[o][2], dm[0], dm[1], dm[2], dm[3]);
        hipLaunchKernelGGL(updateMurBoundary, grid, block, 0, 0, de[t][0], de[t][1], de[t][2], de[s][0], de[s][1],
                           de[s][2], drp[0], drp[1], drp[2], drp[3], drp[4], drp[5]);
    }
    if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) return -2;
    for (int k = 0; k < 3; ++k) ok &= hipMemcpy(h + k * n, dh[k], n * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess;
    for (int b = 0; b < 3; ++b)
        for (int k = 0; k < 3; ++k) {
            ok &= hipMemcpy(e + ((size_t)b * 3 + k) * n, de[b][k], n * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess;
            ok &= hipMemcpy(d + ((size_t)b * 3 + k) * n, dd[b][k], n * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess;
        }
    return ok ? 0 : -3;
}

// Timing of the same launches on a grid of the given extents (vacuum materials, rp = 1, one source
// at the centre, fields at rest): milliseconds per step over `nsteps` steps after `warm` steps, HIP
// events on the null stream.  What the reference's kernels reach on this GPU -- a number to put
// beside the product's, not a parity check.
extern "C" float fdtd_ref_time(const int *dims, const float *consts, int jz_count, const float *jz, int warm, int nsteps)
{
    const int nx = dims[0], ny = dims[1], nz = dims[2];
    const size_t n = (size_t)nx * ny * nz;
    int src[3] = {nx / 2, ny / 2, nz / 2};
    FdtdParams p;
    std::memset(&p, 0, sizeof p);
    p.nx = nx;
    p.ny = ny;
    p.nz = nz;
    p.dt = consts[0];
    p.dx = consts[1];
    p.dy = consts[2];
    p.dz = consts[3];
    p.mu0 = consts[4];
    p.eps0 = consts[5];
    p.sourcesCount = 1;
    p.sources = src;
    p.jzCount = jz_count;
    p.jz = const_cast<float *>(jz);
    copySymbolsToDevice(&p);
    DeviceArrays dev;
    auto alloc = [&](size_t count, float value) -> float * {
        float *ptr = nullptr;
        if (hipMalloc((void **)&ptr, (count + 1024) * sizeof(float)) != hipSuccess) return nullptr;
        dev.all.push_back(ptr);
        std::vector<float> host(std::min<size_t>(count + 1024, (size_t)1 << 22), value);
        for (size_t o = 0; o < count + 1024; o += host.size())
            (void)hipMemcpy(ptr + o, host.data(), std::min(host.size(), count + 1024 - o) * sizeof(float), hipMemcpyHostToDevice);
        return ptr;
    };
    float *dh[3], *de[3][3], *dd[3][3], *dm[4], *drp[6];
    bool ok = true;
    for (int k = 0; k < 3; ++k) ok &= (dh[k] = alloc(n, 0.f)) != nullptr;
    for (int b = 0; b < 3; ++b)
        for (int k = 0; k < 3; ++k) {
            ok &= (de[b][k] = alloc(n, 0.f)) != nullptr;
            ok &= (dd[b][k] = alloc(n, 0.f)) != nullptr;
        }
    const float matv[4] = {0.f, 1.f, 1.f, 0.f}; // sigma, epsI, epsS, tauD of vacuum
    for (int k = 0; k < 4; ++k) ok &= (dm[k] = alloc(n, matv[k])) != nullptr;
    const size_t rpn[6] = {(size_t)2 * ny * nz, (size_t)2 * nx * nz, (size_t)2 * nx * ny,
                           (size_t)2 * ny * nz, (size_t)2 * nx * nz, (size_t)2 * nx * ny};
    for (int f = 0; f < 6; ++f) ok &= (drp[f] = alloc(rpn[f], 1.f)) != nullptr;
    if (!ok) return -1.f;
    const dim3 block(128, 1, 1);
    const dim3 grid((nx + 128 - 1) / 128, ny, nz);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int it = 0; it < warm + nsteps; ++it) {
        if (it == warm) (void)hipEventRecord(e0, 0);
        const int t = it % 3, s = (t + 2) % 3, o = (t + 1) % 3;
        hipLaunchKernelGGL(updateHField, grid, block, 0, 0, dh[0], dh[1], dh[2], de[s][0], de[s][1], de[s][2]);
        hipLaunchKernelGGL(updateDField, grid, block, 0, 0, dd[t][0], dd[t][1], dd[t][2], dd[s][0], dd[s][1], dd[s][2],
                           dh[0], dh[1], dh[2]);
        hipLaunchKernelGGL(updateSources, grid, block, 0, 0, dd[t][2], dd[s][2], dh[0], dh[1], it - it % 3);
        hipLaunchKernelGGL(updateEField, grid, block, 0, 0, de[t][0], de[t][1], de[t][2], de[s][0], de[s][1], de[s][2],
                           de[o][0], de[o][1], de[o][2], dd[t][0], dd[t][1], dd[t][2], dd[s][0], dd[s][1], dd[s][2],
                           dd[o][0], dd[o][1], dd[o][2], dm[0], dm[1], dm[2], dm[3]);
        hipLaunchKernelGGL(updateMurBoundary, grid, block, 0, 0, de[t][0], de[t][1], de[t][2], de[s][0], de[s][1],
                           de[s][2], drp[0], drp[1], drp[2], drp[3], drp[4], drp[5]);
    }
    (void)hipEventRecord(e1, 0);
    float ms = -1.f;
    if (hipEventSynchronize(e1) == hipSuccess && hipGetLastError() == hipSuccess) (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return ms < 0.f ? ms : ms / nsteps;
}
