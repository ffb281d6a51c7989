// HIP kernels of the FDTD field-update path for gfx950 (MI355X).
//
// What each kernel replaces in the reference (UnalignedByte/fdtd):
//   fdtd2d_h, fdtd2d_de, fdtd3d_h, fdtd3d_de : the same equations as an H
//                  kernel and a fused D+source+E kernel
//   fdtd2d_mur, fdtd3d_mur : updateMurBoundary (fdtd_calculations.cu:216-428)
//
// Arithmetic keeps the reference's evaluation order and its fp32/fp64 mix and
// is compiled with -ffp-contract=off, so results are bit-comparable with a
// host build of the same equations.  The kernels are bandwidth bound (about
// 0.5 flop per byte): no MFMA, wide coalesced loads, neighbours through lane
// shuffles.  The fused 2-D sweep lives in sweep2d.hip.
#include <algorithm>

#include "device_math.h"

namespace fdtd {

namespace {

__device__ __forceinline__ bool is_source2d(const SourceList &s, int x, int gy)
{
    bool hit = false;
    for (int i = 0; i < s.count; ++i) hit |= (s.x[i] == x) & (s.y[i] == gy);
    return hit;
}

__device__ __forceinline__ bool is_source3d(const SourceList &s, int x, int y, int gz)
{
    bool hit = false;
    for (int i = 0; i < s.count; ++i) hit |= (s.x[i] == x) & (s.y[i] == y) & (s.z[i] == gz);
    return hit;
}

// ------------------------------------------------------------------ 2-D split

// H update in place, one thread per cell.  blockIdx.y walks storage rows
// starting at the lower halo row when a neighbour slab exists: its hx row is
// recomputed here every step so that only ez has to be exchanged.
__global__ __launch_bounds__(256) void fdtd2d_h(const Sweep2D a)
{
    const Grid2D &g = a.g;
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int j = (int)blockIdx.y - g.has_lower;
    if (x >= g.nx) return;
    const int y = g.gy0 + j;
    const size_t o = (size_t)(j + 1) * g.pitch + x;
    if (x >= 1 && x <= g.nx - 2 && y >= 0 && y <= g.gny - 2)
        a.hx_w[o] = a.hx_r[o] - g.kh_y * (a.ez_n[o + g.pitch] - a.ez_n[o]);
    if (j >= 0 && x <= g.nx - 2 && y >= 1 && y <= g.gny - 2)
        a.hy_w[o] = a.hy_r[o] + g.kh_x * (a.ez_n[o + 1] - a.ez_n[o]);
}

// D + source + E, one thread per cell of the owned rows.
__global__ __launch_bounds__(256) void fdtd2d_de(const Sweep2D a)
{
    const Grid2D &g = a.g;
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int j = blockIdx.y;
    if (x >= g.nx) return;
    const int y = g.gy0 + j;
    const size_t o = (size_t)(j + 1) * g.pitch + x;
    const float d0 = a.dz_n[o];
    const float d1 = a.dz_w[o];
    float dn = d1;
    const bool guard = x >= 1 && x <= g.nx - 2 && y >= 1 && y <= g.gny - 2;
    const bool src = is_source2d(a.src, x, y);
    if (guard || src) {
        const float v = (d0 + g.cx * (a.hy_r[o] - a.hy_r[o - 1])) - g.cy * (a.hx_r[o] - a.hx_r[o - g.pitch]);
        dn = src ? v - a.src.jz : v;
        a.dz_w[o] = dn;
    }
    if (x >= 1) {
        const MatEntry m = a.table[a.code[(size_t)(j + 1) * g.cpitch + x]];
        a.ez_w[o] = e_update(m, g.dt, a.ez_n[o], a.ez_w[o], dn, d0, d1);
    }
}

// ez on the four edges, fdtd_calculations.cu:369-427 with iz = 0.
__global__ __launch_bounds__(256) void fdtd2d_mur(const Mur2D a)
{
    const Grid2D &g = a.g;
    long t = (long)blockIdx.x * 256 + threadIdx.x;
    const size_t pitch = (size_t)g.pitch;
    if (t < 2L * g.nyl) {
        const int side = (int)(t / g.nyl), j = (int)(t % g.nyl);
        const int y = g.gy0 + j;
        if (y < 1 || y > g.gny - 2 || j < a.j_begin || j >= a.j_end) return;
        const size_t row = (size_t)(j + 1) * pitch;
        if (side == 0) {
            const size_t f = row, in = row + 1;
            a.ez_t[f] = mur_face(a.dt, a.dx, a.mu0, a.eps0, a.rpx0[(size_t)y * 2], a.rpx0[(size_t)y * 2 + 1],
                                 a.ez_t[in], a.ez_s[in], a.ez_s[f]);
        } else {
            const size_t f = row + g.nx - 1, in = row + g.nx - 2;
            a.ez_t[f] = mur_face(a.dt, a.dx, a.mu0, a.eps0, a.rpxe[(size_t)y * 2 + 1], a.rpxe[(size_t)y * 2],
                                 a.ez_t[in], a.ez_s[in], a.ez_s[f]);
        }
        return;
    }
    t -= 2L * g.nyl;
    if (t >= 2L * g.nx) return;
    const int side = (int)(t / g.nx), x = (int)(t % g.nx);
    if (x < 1 || x > g.nx - 2) return;
    if (side == 0) {
        if (g.gy0 != 0 || a.j_begin > 0 || a.j_end < 1) return;
        const size_t f = pitch + x, in = 2 * pitch + x;
        a.ez_t[f] = mur_face(a.dt, a.dy, a.mu0, a.eps0, a.rpy0[x], a.rpy0[g.nx + x], a.ez_t[in], a.ez_s[in],
                             a.ez_s[f]);
    } else {
        if (g.gy0 + g.nyl != g.gny || a.j_begin > g.nyl - 1 || a.j_end < g.nyl) return;
        const size_t f = (size_t)g.nyl * pitch + x, in = (size_t)(g.nyl - 1) * pitch + x;
        a.ez_t[f] = mur_face(a.dt, a.dy, a.mu0, a.eps0, a.rpye[g.nx + x], a.rpye[x], a.ez_t[in], a.ez_s[in],
                             a.ez_s[f]);
    }
}

// ------------------------------------------------------------------ 3-D split

// updateHField, fdtd_calculations.cu:32-77 (in place).  blockIdx.z starts at
// the lower halo plane when a neighbour slab exists (hx, hy only).
__global__ __launch_bounds__(256) void fdtd3d_h(const Sweep3D a)
{
    const Grid3D &g = a.g;
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    const int k = (int)blockIdx.z - g.has_lower;
    if (x >= g.nx) return;
    const int z = g.gz0 + k;
    const size_t rs = (size_t)g.pitch, ps = (size_t)g.ny * g.pitch;
    const size_t o = (size_t)(k + 1) * ps + (size_t)y * rs + x;
    const float *ex = a.e_n[0], *ey = a.e_n[1], *ez = a.e_n[2];
    if (x >= 1 && x <= g.nx - 2 && y <= g.ny - 2 && z >= 0 && z <= g.gnz - 2)
        a.h[0][o] = a.h[0][o] - g.kh_y * (ez[o + rs] - ez[o]) + g.kh_y * (ey[o + ps] - ey[o]);
    if (x <= g.nx - 2 && y >= 1 && y <= g.ny - 2 && z >= 0 && z <= g.gnz - 2)
        a.h[1][o] = a.h[1][o] - g.kh_z * (ex[o + ps] - ex[o]) + g.kh_x * (ez[o + 1] - ez[o]);
    if (k >= 0 && x <= g.nx - 2 && y <= g.ny - 2 && z >= 1 && z <= g.gnz - 2)
        a.h[2][o] = a.h[2][o] - g.kh_x * (ey[o + 1] - ey[o]) + g.kh_y * (ex[o + rs] - ex[o]);
}

// updateDField + updateSources + updateEField, fdtd_calculations.cu:80-213.
__global__ __launch_bounds__(256) void fdtd3d_de(const Sweep3D a)
{
    const Grid3D &g = a.g;
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    const int k = blockIdx.z;
    if (x >= g.nx) return;
    const int z = g.gz0 + k;
    const size_t rs = (size_t)g.pitch, ps = (size_t)g.ny * g.pitch;
    const size_t o = (size_t)(k + 1) * ps + (size_t)y * rs + x;
    const float *hx = a.h[0], *hy = a.h[1], *hz = a.h[2];

    const bool x_lo = x >= 1, x_in = x <= g.nx - 2;
    const bool y_lo = y >= 1, y_in = y <= g.ny - 2;
    const bool z_lo = z >= 1, z_in = z <= g.gnz - 2;

    float d0[3], d1[3], dn[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        d0[c] = a.d_n[c][o];
        d1[c] = a.d_w[c][o];
        dn[c] = d1[c];
    }
    if (x_in && y_lo && y_in && z_lo && z_in) {
        dn[0] = (d0[0] + g.cy * (hz[o] - hz[o - rs])) - g.cz * (hy[o] - hy[o - ps]);
        a.d_w[0][o] = dn[0];
    }
    if (x_lo && x_in && y_in && z_lo && z_in) {
        dn[1] = (d0[1] + g.cz * (hx[o] - hx[o - ps])) - g.cx * (hz[o] - hz[o - 1]);
        a.d_w[1][o] = dn[1];
    }
    const bool src = is_source3d(a.src, x, y, z);
    if ((x_lo && x_in && y_lo && y_in && z_in) || src) {
        const float v = (d0[2] + g.cx * (hy[o] - hy[o - 1])) - g.cy * (hx[o] - hx[o - rs]);
        dn[2] = src ? v - a.src.jz : v;
        a.d_w[2][o] = dn[2];
    }

    const bool g_ex = x_in && y_lo && y_in && z_lo && z_in;
    const bool g_eyz = x_lo && z_lo;
    if (g_ex || g_eyz) {
        const MatEntry m = a.table[a.code[((size_t)(k + 1) * g.ny + y) * g.rowlen + x]];
        if (g_ex) a.e_w[0][o] = e_update(m, g.dt, a.e_n[0][o], a.e_w[0][o], dn[0], d0[0], d1[0]);
        if (g_eyz) {
            a.e_w[1][o] = e_update(m, g.dt, a.e_n[1][o], a.e_w[1][o], dn[1], d0[1], d1[1]);
            a.e_w[2][o] = e_update(m, g.dt, a.e_n[2][o], a.e_w[2][o], dn[2], d0[2], d1[2]);
        }
    }
}

// updateMurBoundary, fdtd_calculations.cu:216-428: one thread per face cell,
// the two components that face carries.
__global__ __launch_bounds__(256) void fdtd3d_mur(const Mur3D a)
{
    const Grid3D &g = a.g;
    const long nxf = 2L * g.ny * g.nzl, nyf = 2L * g.nx * g.nzl, nzf = 2L * g.nx * g.ny;
    long t = (long)blockIdx.x * 256 + threadIdx.x;
    const size_t rs = (size_t)g.pitch, ps = (size_t)g.ny * g.pitch;
    float *ext = a.e_t[0], *eyt = a.e_t[1], *ezt = a.e_t[2];
    const float *exs = a.e_s[0], *eys = a.e_s[1], *ezs = a.e_s[2];
    const float dt = a.dt, mu0 = a.mu0, eps0 = a.eps0;

    if (t < nxf) { // x faces: ey (:303-329) and ez (:369-395)
        const int side = (int)(t / ((long)g.ny * g.nzl));
        const long r = t % ((long)g.ny * g.nzl);
        const int k = (int)(r / g.ny), y = (int)(r % g.ny);
        const int z = g.gz0 + k;
        const size_t base = (size_t)(k + 1) * ps + (size_t)y * rs;
        const float *rp = side ? a.rp[FDTD_RP_XEND] : a.rp[FDTD_RP_X0];
        const size_t ri = ((size_t)z * g.ny + y) * 2;
        const float r_here = side ? rp[ri + 1] : rp[ri], r_in = side ? rp[ri] : rp[ri + 1];
        const size_t f = base + (side ? g.nx - 1 : 0), in = base + (side ? g.nx - 2 : 1);
        if (k < a.k_begin || k >= a.k_end) return;
        if (y <= g.ny - 2 && z >= 1 && z <= g.gnz - 2) {
            // the high face takes eySource for both interior terms (:323, :325)
            const float et_in = side ? eys[in] : eyt[in];
            eyt[f] = mur_face(dt, a.dx, mu0, eps0, r_here, r_in, et_in, eys[in], eys[f]);
        }
        if (y >= 1 && y <= g.ny - 2 && z <= g.gnz - 2)
            ezt[f] = mur_face(dt, a.dx, mu0, eps0, r_here, r_in, ezt[in], ezs[in], ezs[f]);
        return;
    }
    t -= nxf;
    if (t < nyf) { // y faces: ex (:237-263) and ez (:401-427)
        const int side = (int)(t / ((long)g.nx * g.nzl));
        const long r = t % ((long)g.nx * g.nzl);
        const int k = (int)(r / g.nx), x = (int)(r % g.nx);
        const int z = g.gz0 + k;
        const size_t base = (size_t)(k + 1) * ps + x;
        const float *rp = side ? a.rp[FDTD_RP_YEND] : a.rp[FDTD_RP_Y0];
        const size_t ri = (size_t)z * 2 * g.nx + x;
        const float r_here = side ? rp[ri + g.nx] : rp[ri], r_in = side ? rp[ri] : rp[ri + g.nx];
        const size_t f = base + (side ? (size_t)(g.ny - 1) * rs : 0);
        const size_t in = base + (side ? (size_t)(g.ny - 2) * rs : rs);
        if (k < a.k_begin || k >= a.k_end) return;
        if (x <= g.nx - 2 && z >= 1 && z <= g.gnz - 2)
            ext[f] = mur_face(dt, a.dy, mu0, eps0, r_here, r_in, ext[in], exs[in], exs[f]);
        if (x >= 1 && x <= g.nx - 2 && z <= g.gnz - 2)
            ezt[f] = mur_face(dt, a.dy, mu0, eps0, r_here, r_in, ezt[in], ezs[in], ezs[f]);
        return;
    }
    t -= nyf;
    if (t < nzf) { // z faces: ex (:269-295) and ey (:335-361); first and last slab only
        const int side = (int)(t / ((long)g.nx * g.ny));
        const long r = t % ((long)g.nx * g.ny);
        const int y = (int)(r / g.nx), x = (int)(r % g.nx);
        if (side == 0 ? g.gz0 != 0 : g.gz0 + g.nzl != g.gnz) return;
        const int k = side ? g.nzl - 1 : 0, kin = side ? g.nzl - 2 : 1;
        if (k < a.k_begin || k >= a.k_end) return;
        const float *rp = side ? a.rp[FDTD_RP_ZEND] : a.rp[FDTD_RP_Z0];
        const size_t ri = (size_t)y * g.nx + x, rl = (size_t)g.ny * g.nx;
        const float r_here = side ? rp[ri + rl] : rp[ri], r_in = side ? rp[ri] : rp[ri + rl];
        const size_t f = (size_t)(k + 1) * ps + (size_t)y * rs + x;
        const size_t in = (size_t)(kin + 1) * ps + (size_t)y * rs + x;
        if (x <= g.nx - 2 && y >= 1 && y <= g.ny - 2)
            ext[f] = mur_face(dt, a.dz, mu0, eps0, r_here, r_in, ext[in], exs[in], exs[f]);
        if (x >= 1 && x <= g.nx - 2 && y <= g.ny - 2)
            eyt[f] = mur_face(dt, a.dz, mu0, eps0, r_here, r_in, eyt[in], eys[in], eys[f]);
    }
}

__global__ __launch_bounds__(256) void fdtd3d_mur_coefs(const MurCoefs a)
{
    const int axis = a.face % 3;
    const bool high = a.face >= 3;
    const long t = a.first + (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= a.first + a.count) return;
    float r0, r1; // the two rp entries of the cell pair, lower index first (layouts of fdtd.cu:952-1001)
    if (axis == 0) {
        r0 = a.rp[t * 2];
        r1 = a.rp[t * 2 + 1];
    } else if (axis == 1) {
        const long z = t / a.nx, x = t % a.nx;
        r0 = a.rp[z * 2 * a.nx + x];
        r1 = a.rp[z * 2 * a.nx + x + a.nx];
    } else {
        r0 = a.rp[t];
        r1 = a.rp[t + (long)a.nx * a.ny];
    }
    ((f4 *)a.cf)[t] = mur_coefs(a.dt, a.du, a.mu0, a.eps0, high ? r1 : r0, high ? r0 : r1);
}

// ------------------------------------------------------------------ probes

__global__ void fdtd2d_probe(const Probe2D a)
{
    const int i = threadIdx.x;
    if (i >= a.p.count) return;
    const int j = a.p.y[i] - a.g.gy0;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (j >= 0 && j < a.g.nyl && a.p.x[i] >= 0 && a.p.x[i] < a.g.nx) {
        const size_t o = (size_t)(j + 1) * a.g.pitch + a.p.x[i];
        v[0] = a.hx[o];
        v[1] = a.hy[o];
        v[2] = a.dz[o];
        v[3] = a.ez[o];
    }
    else {
        // not on this slab: the multi-step sweep wrote nothing for it into the earlier slots
        for (int m = 0; m < a.nmid; ++m) {
            float *mid = a.ring + ((size_t)a.mid_slot[m] * FDTD_MAX_PROBES + i) * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) mid[c] = 0.f;
        }
    }
    float *dst = a.ring + ((size_t)a.slot * FDTD_MAX_PROBES + i) * 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) dst[c] = v[c];
}

// blockIdx.y = (side * 7 + array) * depth + row; one thread per column.
__global__ __launch_bounds__(256) void fdtd2d_halo_deep(const HaloDeep a)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= a.nx) return;
    const int r = blockIdx.y % a.depth;
    const int arr = (blockIdx.y / a.depth) % 7;
    const int side = blockIdx.y / (7 * a.depth);
    if (!((a.sides >> side) & 1)) return;
    // packed: the rows a neighbour needs (first / last `depth` owned rows); unpacked: the halo rows
    const int row = a.unpack ? (side == 0 ? r - a.depth : a.nyl + r) : (side == 0 ? r : a.nyl - a.depth + r);
    const size_t cell = (size_t)(row + 1) * a.pitch + x; // storage row 0 is local row -1
    if (arr < 6) {
        float *slot = a.buf[side] + ((size_t)arr * a.depth + r) * a.nx + x;
        if (a.unpack)
            a.arr[arr][cell] = *slot;
        else
            *slot = a.arr[arr][cell];
    } else {
        uint8_t *slot = (uint8_t *)(a.buf[side] + (size_t)6 * a.depth * a.nx) + (size_t)r * a.nx + x;
        const size_t ccell = (size_t)(row + 1) * a.cpitch + x;
        if (a.unpack)
            a.code[ccell] = *slot;
        else
            *slot = a.code[ccell];
    }
}

// One thread per four cells of a row (one 4-byte store); block id = (plane * rows + row) * xblocks + x block.
__global__ __launch_bounds__(256) void fdtd_expand_codes(const ExpandCodes a, const unsigned xblocks)
{
    const unsigned xb = blockIdx.x % xblocks;
    const long line = blockIdx.x / xblocks;
    const int x4 = (int)(xb * 256 + threadIdx.x) * 4;
    if (x4 >= a.nx) return;
    const int j = (int)(line % a.rows);
    const long k = line / a.rows;
    const uint8_t *src = a.base + ((size_t)k * a.by + (size_t)((a.row0 + j) / a.up)) * a.bx;
    unsigned word = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (x4 + i < a.nx) word |= (unsigned)src[(x4 + i) / a.up] << (8 * i);
    *(unsigned *)(a.code + (size_t)k * a.plane_pitch + (size_t)j * a.row_pitch + x4) = word;
}

__global__ void fdtd3d_probe(const Probe3D a)
{
    const int i = threadIdx.x;
    if (i >= a.p.count) return;
    const int k = a.p.z[i] - a.g.gz0;
    float *dst = a.ring + ((size_t)a.slot * FDTD_MAX_PROBES + i) * 9;
    const bool in = k >= 0 && k < a.g.nzl && a.p.x[i] >= 0 && a.p.x[i] < a.g.nx && a.p.y[i] >= 0 &&
                    a.p.y[i] < a.g.ny;
    const size_t o = in ? ((size_t)(k + 1) * a.g.ny + a.p.y[i]) * a.g.pitch + a.p.x[i] : 0;
#pragma unroll
    for (int c = 0; c < 9; ++c) dst[c] = in ? a.f[c][o] : 0.f;
}

} // namespace

// ------------------------------------------------------------------ launchers
// Every launcher checks on the host that the grid it builds matches what the
// kernel indexes; the callers guarantee the array extents (context.hip).

static inline hipStream_t S(void *s) { return (hipStream_t)s; }

void launch_h2d(const Sweep2D &a, void *stream)
{
    dim3 grid((a.g.nx + 255) / 256, a.g.nyl + a.g.has_lower, 1);
    hipLaunchKernelGGL(fdtd2d_h, grid, dim3(256), 0, S(stream), a);
}

void launch_de2d(const Sweep2D &a, void *stream)
{
    dim3 grid((a.g.nx + 255) / 256, a.g.nyl, 1);
    hipLaunchKernelGGL(fdtd2d_de, grid, dim3(256), 0, S(stream), a);
}

void launch_mur2d(const Mur2D &a, void *stream)
{
    const long total = 2L * a.g.nyl + 2L * a.g.nx;
    hipLaunchKernelGGL(fdtd2d_mur, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, S(stream), a);
}

void launch_h3d(const Sweep3D &a, void *stream)
{
    dim3 grid((a.g.nx + 255) / 256, a.g.ny, a.g.nzl + a.g.has_lower);
    hipLaunchKernelGGL(fdtd3d_h, grid, dim3(256), 0, S(stream), a);
}

void launch_de3d(const Sweep3D &a, void *stream)
{
    dim3 grid((a.g.nx + 255) / 256, a.g.ny, a.g.nzl);
    hipLaunchKernelGGL(fdtd3d_de, grid, dim3(256), 0, S(stream), a);
}

void launch_mur3d(const Mur3D &a, void *stream)
{
    const long total = 2L * a.g.ny * a.g.nzl + 2L * a.g.nx * a.g.nzl + 2L * a.g.nx * a.g.ny;
    hipLaunchKernelGGL(fdtd3d_mur, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, S(stream), a);
}

void launch_halo_deep(const HaloDeep &a, void *stream)
{
    hipLaunchKernelGGL(fdtd2d_halo_deep, dim3((a.nx + 255) / 256, 2 * 7 * a.depth), dim3(256), 0, S(stream), a);
}

void launch_mur_coefs(const MurCoefs &a, void *stream)
{
    if (a.count <= 0) return;
    hipLaunchKernelGGL(fdtd3d_mur_coefs, dim3((unsigned)((a.count + 255) / 256)), dim3(256), 0, S(stream), a);
}

void launch_expand_codes(const ExpandCodes &a, void *stream)
{
    const unsigned xblocks = (unsigned)((a.nx + 1023) / 1024);
    // at most 2^31 blocks per launch: planes go in groups
    const long lines_per_launch = std::max<long>(a.rows, (long)(0x7fffffffL / xblocks) / a.rows * a.rows);
    const long total = (long)a.rows * a.planes;
    for (long first = 0; first < total; first += lines_per_launch) {
        ExpandCodes b = a;
        const long lines = std::min(lines_per_launch, total - first);
        b.base = a.base + (size_t)(first / a.rows) * a.by * a.bx;
        b.code = a.code + (size_t)(first / a.rows) * a.plane_pitch;
        b.planes = (int)(lines / a.rows);
        hipLaunchKernelGGL(fdtd_expand_codes, dim3((unsigned)(lines * xblocks)), dim3(256), 0, S(stream), b, xblocks);
    }
}

void launch_probe2d(const Probe2D &a, void *stream)
{
    hipLaunchKernelGGL(fdtd2d_probe, dim3(1), dim3(64), 0, S(stream), a);
}

void launch_probe3d(const Probe3D &a, void *stream)
{
    hipLaunchKernelGGL(fdtd3d_probe, dim3(1), dim3(64), 0, S(stream), a);
}

const char *de2d_kernel_name() { return "fdtd2d_de"; }
const char *de3d_kernel_name() { return "fdtd3d_de"; }

} // namespace fdtd
