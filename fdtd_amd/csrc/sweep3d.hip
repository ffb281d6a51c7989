// fdtd3d_sweep: one pass over memory per time step of the 3-D grid --
// updateHField + updateDField + updateSources + updateEField
// (fdtd_calculations.cu:32-213) fused.
//
// Work decomposition.  A plane of a field is one flat run of 16-byte words (rows
// of pitch/4 words, ny rows); a thread owns one word, i.e. four consecutive x
// cells of one (y, z) row, and a block owns 256 consecutive words -- 4 KB of
// every array.  Each plane is cut into 8 segments, one per XCD; an XCD walks its
// segment plane after plane, so what it re-reads of the planes z-1 and z+1 is
// still in its own L2.
//
// Neighbours.  D^{n+1}(c) needs the NEW H at c and at c-x, c-y, c-z.  Those six
// neighbour values are recomputed here from the old H and E^n of the
// neighbouring rows (lines the blocks before this one have just pulled into
// L2 / the Infinity Cache), the x neighbours come from the adjacent lane, and
// the lanes at the ends of a wave fetch the few scalars they lack.  H is read
// from one generation and written to the other, so a recomputed value never
// races with its owner's store.
//
// Traffic per cell from HBM: 3 H + 3 E^n + 3 E^{n-1} + 3 D^n + 3 D^{n-1} in
// (60 B) + 1 B material code, 3 H + 3 E^{n+1} + 3 D^{n+1} out (36 B) = 97 B.
#include <cstdlib>

#include "device_math.h"

namespace fdtd {

namespace {

__device__ __forceinline__ f4 ld4(const float *p, bool act, bool stream)
{
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    if (!act) return zero;
    return stream ? ld4_stream(p) : *(const f4 *)p;
}

// {v[1], v[2], v[3], first element of the next lane (or `edge` on lane 63)}
__device__ __forceinline__ f4 shift_left(f4 v, int lane, float edge)
{
    float nxt = __shfl_down(v.x, 1);
    if (lane == 63) nxt = edge;
    const f4 r = {v.y, v.z, v.w, nxt};
    return r;
}

// {last element of the previous lane (or `edge` on lane 0), v[0], v[1], v[2]}
__device__ __forceinline__ f4 shift_right(f4 v, int lane, float edge)
{
    float prv = __shfl_up(v.w, 1);
    if (lane == 0) prv = edge;
    const f4 r = {prv, v.x, v.y, v.z};
    return r;
}

#ifndef FDTD_3D_LDS_TABLE
#define FDTD_3D_LDS_TABLE 1
#endif
constexpr bool LDS_TABLE = FDTD_3D_LDS_TABLE != 0;

// FLAT: the segment space is all planes of the launch (small grids), not one plane.
template <bool ONE, int W, bool MUR, bool FLAT> __global__ __launch_bounds__(256, W) void fdtd3d_sweep(const Fused3D a)
{
    const Grid3D &g = a.g;
    const int lane = threadIdx.x & 63;
    if (a.ring && blockIdx.x == gridDim.x - 1) {
        // the extra workgroup: the probe row of the previous step, fdtd3d_probe's layout
        const int i = threadIdx.x;
        if (i < a.probes.count) {
            const int kp = a.probes.z[i] - g.gz0;
            float *dst = a.ring + ((size_t)a.probe_slot * FDTD_MAX_PROBES + i) * 9;
            const bool in = kp >= 0 && kp < g.nzl && a.probes.x[i] >= 0 && a.probes.x[i] < g.nx && a.probes.y[i] >= 0 &&
                            a.probes.y[i] < g.ny;
            const size_t po = in ? ((size_t)(kp + 1) * g.ny + a.probes.y[i]) * g.pitch + a.probes.x[i] : 0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                dst[c] = in ? a.h_r[c][po] : 0.f;
                dst[3 + c] = in ? a.d_n[c][po] : 0.f;
                dst[6 + c] = in ? a.e_n[c][po] : 0.f;
            }
        }
        return;
    }
    // Grids with several materials: the coefficient table goes into LDS (4 KB per workgroup, one row
    // per thread, asked for before every other load; the barrier sits in front of the E update).  On
    // the body model three waves of five hold more than one tissue -- a row of 265 cells that crosses
    // the body -- and gathered their four rows per lane from memory at the very end of the thread.
    __shared__ MatEntry tab[ONE ? 1 : FDTD_MAX_CODES];
    if (!ONE && LDS_TABLE)
        for (int i = threadIdx.x; i < FDTD_MAX_CODES; i += blockDim.x) tab[i] = a.table[i];
    // Block ids are dealt to the 8 XCDs in turn.  XCD i (block id mod 8) owns segment i --
    // one eighth of the words -- of EVERY plane and walks the planes in z order, so the
    // lines of plane z-1 and z+1 it needs are the ones it pulled into its own L2 a plane ago.
    const int p4 = g.rowlen >> 2;
    const unsigned xcd = blockIdx.x & 7u, seq = blockIdx.x >> 3;
    // A segment space is one plane, or -- small grids, whose planes would leave a fifth of the lanes
    // of wave-aligned segments idle -- the `flat` planes of the launch taken as one run of words.
    const unsigned in_seg = (seq % (unsigned)a.seg_blocks) * blockDim.x + threadIdx.x;
    const unsigned in_space = xcd * (unsigned)a.seg_words + in_seg;      // word within the segment space
    const unsigned row = in_space / (unsigned)p4;
    const int x4 = (int)(in_space - row * (unsigned)p4) * 4;
    const unsigned kp = FLAT ? row / (unsigned)g.ny : 0u;
    const int y = (int)(row - kp * (unsigned)g.ny);
    const int k = a.k_begin + (FLAT ? (int)kp : (int)(seq / (unsigned)a.seg_blocks));
    const long long in_plane = (long long)in_space - (long long)kp * a.plane_words;
    const long long t = (long long)k * a.plane_words + in_plane;
    const bool act = in_seg < (unsigned)a.seg_words &&
                     (FLAT ? (long long)in_space < (long long)a.flat * a.plane_words : in_space < (unsigned)a.plane_words) &&
                     k < a.k_end && x4 < g.nx;
    const int z = g.gz0 + k;
    const size_t rs = (size_t)g.pitch, ps = (size_t)g.ny * g.pitch;
    const size_t o = act ? (size_t)(k + 1) * ps + (size_t)y * rs + x4 : ps;
    const size_t co = act ? ((size_t)(k + 1) * g.ny + (size_t)y) * g.rowlen + x4 : 0; // code array: rows of rowlen bytes

    const float *ex = a.e_n[0], *ey = a.e_n[1], *ez = a.e_n[2];
    const float *hx = a.h_r[0], *hy = a.h_r[1], *hz = a.h_r[2];

    // Material codes first, and for a wave that holds one material the table row right behind
    // them: a chain of two dependent loads that would otherwise start only after the field loads
    // have come back.  (A mixed wave gathers its rows next to the E update: they stay in cache.)
    const unsigned cd = act ? *(const unsigned *)(a.code + co) : 0u;
    bool uniform = true;
    MatEntry mu = a.one;
    if (!ONE) {
        const unsigned c0 = (unsigned)__builtin_amdgcn_readfirstlane((int)cd) & 0xffu;
        uniform = __all(cd == c0 * 0x01010101u);
        if (uniform) mu = a.table[c0];
    }

    // Mur factors of the x faces next to this word: every wave holds such words, so they are asked for
    // here, ahead of the field loads, not behind the E update
    int hi_cell = -1; // which of the four cells is nx-2
    f4 cfx0 = {0.f, 0.f, 0.f, 0.f}, cfxe = cfx0;
    if (MUR && act) {
        const size_t fx = (size_t)(g.gz0 + k) * g.ny + y;
        if (x4 == 0) cfx0 = ((const f4 *)a.cf[FDTD_RP_X0])[fx];
        if (g.nx - 2 >= x4 && g.nx - 2 < x4 + 4) {
            hi_cell = g.nx - 2 - x4;
            cfxe = ((const f4 *)a.cf[FDTD_RP_XEND])[fx];
        }
    }
    // the same for the y and z faces (whole rows and planes of words; rows 1 and ny-2, planes 1 and
    // gnz-2 are distinct: ny, nz >= 4)
    f4 cfy[4], cfz[4];
    if (MUR) {
        const int zz = g.gz0 + k;
        const bool ynear = act && (y == 1 || y == g.ny - 2), znear = act && (zz == 1 || zz == g.gnz - 2);
        const f4 *py = (const f4 *)(y == 1 ? a.cf[FDTD_RP_Y0] : a.cf[FDTD_RP_YEND]) + (size_t)zz * g.nx;
        const f4 *pz = (const f4 *)(zz == 1 ? a.cf[FDTD_RP_Z0] : a.cf[FDTD_RP_ZEND]) + (size_t)y * g.nx;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f4 zero = {0.f, 0.f, 0.f, 0.f};
            cfy[i] = (ynear && x4 + i < g.nx) ? py[x4 + i] : zero;
            cfz[i] = (znear && x4 + i < g.nx) ? pz[x4 + i] : zero;
        }
    }

    // E^n and old H: own row and the rows below in y and z (and the two diagonals)
    const f4 ex_c = ld4(ex + o, act, false), ex_py = ld4(ex + o + rs, act, false), ex_pz = ld4(ex + o + ps, act, false);
    const f4 ex_my = ld4(ex + o - rs, act, false), ex_mz = ld4(ex + o - ps, act, true);
    const f4 ey_c = ld4(ey + o, act, false), ey_pz = ld4(ey + o + ps, act, false), ey_my = ld4(ey + o - rs, act, false);
    const f4 ey_mypz = ld4(ey + o - rs + ps, act, false), ey_mz = ld4(ey + o - ps, act, true);
    const f4 ez_c = ld4(ez + o, act, false), ez_py = ld4(ez + o + rs, act, false), ez_my = ld4(ez + o - rs, act, false);
    const f4 ez_mz = ld4(ez + o - ps, act, true), ez_mzpy = ld4(ez + o - ps + rs, act, false);
    const f4 hx_c = ld4(hx + o, act, false), hx_my = ld4(hx + o - rs, act, false), hx_mz = ld4(hx + o - ps, act, true);
    const f4 hy_c = ld4(hy + o, act, false), hy_mz = ld4(hy + o - ps, act, true);
    const f4 hz_c = ld4(hz + o, act, false), hz_my = ld4(hz + o - rs, act, true);

    // scalars the lanes at the two ends of the wave lack
    const bool edge_r = (lane == 63) && act && (x4 + 4 < g.nx);
    const bool edge_l = (lane == 0) && act && (x4 > 0);
    float r_ez = 0.f, r_ey = 0.f, r_ey_my = 0.f, r_ez_mz = 0.f;
    if (edge_r) {
        r_ez = ez[o + 4];
        r_ey = ey[o + 4];
        r_ey_my = ey[o - rs + 4];
        r_ez_mz = ez[o - ps + 4];
    }
    float l_hy = 0.f, l_hz = 0.f, l_ex = 0.f, l_ex_pz = 0.f, l_ex_py = 0.f, l_ez = 0.f, l_ey = 0.f;
    if (edge_l) {
        l_hy = hy[o - 1];
        l_hz = hz[o - 1];
        l_ex = ex[o - 1];
        l_ex_pz = ex[o - 1 + ps];
        l_ex_py = ex[o - 1 + rs];
        l_ez = ez[o - 1];
        l_ey = ey[o - 1];
    }

    // guards
    bool x_lo[4], x_hi[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        x_lo[i] = x4 + i >= 1;
        x_hi[i] = x4 + i <= g.nx - 2;
    }
    const bool y_lo = y >= 1, y_hi = y <= g.ny - 2, z_lo = z >= 1, z_hi = z <= g.gnz - 2;
    const bool ym_lo = y - 1 >= 1, ym_ok = y - 1 >= 0;           // row y-1
    const bool zm_lo = z - 1 >= 1, zm_ok = z - 1 >= 0;           // plane z-1

    const f4 ez_px = shift_left(ez_c, lane, r_ez);
    const f4 ey_px = shift_left(ey_c, lane, r_ey);
    const f4 ey_my_px = shift_left(ey_my, lane, r_ey_my);
    const f4 ez_mz_px = shift_left(ez_mz, lane, r_ez_mz);

    f4 hx_n, hy_n, hz_n, hx_ny, hz_ny, hx_nz, hy_nz;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // own cell, fdtd_calculations.cu:44-76
        hx_n[i] = (x_lo[i] && x_hi[i] && y_hi && z_hi)
                      ? hx_c[i] - g.kh_y * (ez_py[i] - ez_c[i]) + g.kh_y * (ey_pz[i] - ey_c[i]) : hx_c[i];
        hy_n[i] = (x_hi[i] && y_lo && y_hi && z_hi)
                      ? hy_c[i] - g.kh_z * (ex_pz[i] - ex_c[i]) + g.kh_x * (ez_px[i] - ez_c[i]) : hy_c[i];
        hz_n[i] = (x_hi[i] && y_hi && z_lo && z_hi)
                      ? hz_c[i] - g.kh_x * (ey_px[i] - ey_c[i]) + g.kh_y * (ex_py[i] - ex_c[i]) : hz_c[i];
        // row y-1 (its y+1 neighbour is this row): y-1 <= ny-2 always holds
        hx_ny[i] = (x_lo[i] && x_hi[i] && ym_ok && z_hi)
                       ? hx_my[i] - g.kh_y * (ez_c[i] - ez_my[i]) + g.kh_y * (ey_mypz[i] - ey_my[i]) : hx_my[i];
        hz_ny[i] = (x_hi[i] && ym_ok && z_lo && z_hi)
                       ? hz_my[i] - g.kh_x * (ey_my_px[i] - ey_my[i]) + g.kh_y * (ex_c[i] - ex_my[i]) : hz_my[i];
        // plane z-1 (its z+1 neighbour is this plane): z-1 <= gnz-2 always holds
        hx_nz[i] = (x_lo[i] && x_hi[i] && y_hi && zm_ok)
                       ? hx_mz[i] - g.kh_y * (ez_mzpy[i] - ez_mz[i]) + g.kh_y * (ey_c[i] - ey_mz[i]) : hx_mz[i];
        hy_nz[i] = (x_hi[i] && y_lo && y_hi && zm_ok)
                       ? hy_mz[i] - g.kh_z * (ex_c[i] - ex_mz[i]) + g.kh_x * (ez_mz_px[i] - ez_mz[i]) : hy_mz[i];
    }
    (void)ym_lo;
    (void)zm_lo;

    // new hy, hz of the cell left of the wave (x4 - 1 <= nx-2 holds there)
    float e_hy = l_hy, e_hz = l_hz;
    if (edge_l) {
        if (y_lo && y_hi && z_hi) e_hy = l_hy - g.kh_z * (l_ex_pz - l_ex) + g.kh_x * (ez_c.x - l_ez);
        if (y_hi && z_lo && z_hi) e_hz = l_hz - g.kh_x * (ey_c.x - l_ey) + g.kh_y * (l_ex_py - l_ex);
    }
    const f4 hy_nx = shift_right(hy_n, lane, e_hy);
    const f4 hz_nx = shift_right(hz_n, lane, e_hz);

    // The nine pointwise-only words (D^n, D^{n-1}, E^{n-1}) are loaded here, behind the H arithmetic
    // that retires most of the 22 stencil words: fewer registers are live at once, three waves fit
    // a SIMD instead of two and cover each other's waits (512^3: 2.49 ms per sweep against 2.82).
    __builtin_amdgcn_sched_barrier(0);
    f4 d0[3], d1[3], e1[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        d0[c] = ld4(a.d_n[c] + o, act, true);
        d1[c] = ld4(a.d_w[c] + o, act, true);
        e1[c] = ld4(a.e_w[c] + o, act, true);
    }

    // sources of this word
    unsigned srcflag = 0;
    for (int s = 0; s < a.nsrc; ++s)
        if (a.src_word[s] == t) srcflag |= 1u << a.src_comp[s];

    // D, fdtd_calculations.cu:97-121 and :205-210
    f4 dn[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float vx = (d0[0][i] + g.cy * (hz_n[i] - hz_ny[i])) - g.cz * (hy_n[i] - hy_nz[i]);
        const float vy = (d0[1][i] + g.cz * (hx_n[i] - hx_nz[i])) - g.cx * (hz_n[i] - hz_nx[i]);
        const float vz = (d0[2][i] + g.cx * (hy_n[i] - hy_nx[i])) - g.cy * (hx_n[i] - hx_ny[i]);
        dn[0][i] = (x_hi[i] && y_lo && y_hi && z_lo && z_hi) ? vx : d1[0][i];
        dn[1][i] = (x_lo[i] && x_hi[i] && y_hi && z_lo && z_hi) ? vy : d1[1][i];
        float dz = (x_lo[i] && x_hi[i] && y_lo && y_hi && z_hi) ? vz : d1[2][i];
        if ((srcflag >> i) & 1u) dz = vz - a.jz;
        dn[2][i] = dz;
    }

    // E, fdtd_calculations.cu:147-182
    if (!ONE && LDS_TABLE) __syncthreads();
    f4 en[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        MatEntry m = mu;
        if (!ONE && !uniform) m = LDS_TABLE ? tab[(cd >> (8 * i)) & 0xffu] : a.table[(cd >> (8 * i)) & 0xffu];
        const bool g_ex = x_hi[i] && y_lo && y_hi && z_lo && z_hi;
        const bool g_eyz = x_lo[i] && z_lo;
        en[0][i] = g_ex ? e_update(m, g.dt, ex_c[i], e1[0][i], dn[0][i], d0[0][i], d1[0][i]) : e1[0][i];
        en[1][i] = g_eyz ? e_update(m, g.dt, ey_c[i], e1[1][i], dn[1][i], d0[1][i], d1[1][i]) : e1[1][i];
        en[2][i] = g_eyz ? e_update(m, g.dt, ez_c[i], e1[2][i], dn[2][i], d0[2][i], d1[2][i]) : e1[2][i];
    }

    // updateMurBoundary, fdtd_calculations.cu:216-428, for the face cells next to this thread's cells.
    // A face value needs E^{n+1} and E^n of the cell one step inside (this thread's) and E^n of the
    // face cell (a neighbour word loaded for the H update above).  No cell that is one step inside a
    // face is itself finished by a Mur case of the same component, so en[] is final here.
    unsigned skip[3] = {0u, 0u, 0u}; // cells of the own word that another Mur case finishes
    if (MUR && act) {
        const bool y_in = y_lo && y_hi, z_in = z_lo && z_hi;
        const bool y_face = y == 0 || y == g.ny - 1, z_face = z == 0 || z == g.gnz - 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int x = x4 + i;
            const bool x_in = x_lo[i] && x_hi[i], x_face = x == 0 || x == g.nx - 1;
            if ((y_face && x_hi[i] && z_in) || (z_face && x_hi[i] && y_in)) skip[0] |= 1u << i;
            if ((x_face && y_hi && z_in) || (z_face && x_in && y_hi)) skip[1] |= 1u << i;
            if ((x_face && y_in && z_hi) || (y_face && x_in && z_hi)) skip[2] |= 1u << i;
        }
        // x faces: ey (:303-329) and ez (:369-395)
        if (x4 == 0) { // cell 1 is one step inside the low face, cell 0 on it
            if (y_hi && z_in) a.e_w[1][o] = mur_apply(cfx0, en[1][1], ey_c[1], ey_c[0]);
            if (y_in && z_hi) a.e_w[2][o] = mur_apply(cfx0, en[2][1], ez_c[1], ez_c[0]);
        }
        if (hi_cell >= 0) {
            auto pick = [&](const f4 &v) { return hi_cell == 0 ? v.x : hi_cell == 1 ? v.y : hi_cell == 2 ? v.z : v.w; };
            const float ey_in = pick(ey_c), ez_in = pick(ez_c);
            // the high face takes eySource for both interior terms (:323, :325)
            if (y_hi && z_in) a.e_w[1][o + hi_cell + 1] = mur_apply(cfxe, ey_in, ey_in, pick(ey_px));
            if (y_in && z_hi) a.e_w[2][o + hi_cell + 1] = mur_apply(cfxe, pick(en[2]), ez_in, pick(ez_px));
        }
        // y faces: ex (:237-263) and ez (:401-427); rows 1 and ny-2 are distinct (ny >= 4)
        if (y == 1 || y == g.ny - 2) {
            const bool low = y == 1;
            const f4 exf = low ? ex_my : ex_py, ezf = low ? ez_my : ez_py;
            const size_t fo = low ? o - rs : o + rs;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int x = x4 + i;
                if (x >= g.nx) continue;
                const f4 cf = cfy[i];
                if (x_hi[i] && z_in) a.e_w[0][fo + i] = mur_apply(cf, en[0][i], ex_c[i], exf[i]);
                if (x_lo[i] && x_hi[i] && z_hi) a.e_w[2][fo + i] = mur_apply(cf, en[2][i], ez_c[i], ezf[i]);
            }
        }
        // z faces: ex (:269-295) and ey (:335-361); planes 1 and gnz-2 are distinct (nz >= 4)
        if (z == 1 || z == g.gnz - 2) {
            const bool low = z == 1;
            const f4 exf = low ? ex_mz : ex_pz, eyf = low ? ey_mz : ey_pz;
            const size_t fo = low ? o - ps : o + ps;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int x = x4 + i;
                if (x >= g.nx) continue;
                const f4 cf = cfz[i];
                if (x_hi[i] && y_in) a.e_w[0][fo + i] = mur_apply(cf, en[0][i], ex_c[i], exf[i]);
                if (x_lo[i] && x_hi[i] && y_hi) a.e_w[1][fo + i] = mur_apply(cf, en[1][i], ey_c[i], eyf[i]);
            }
        }
    }

    if (act) {
        st4_stream(a.h_w[0] + o, hx_n);
        st4_stream(a.h_w[1] + o, hy_n);
        st4_stream(a.h_w[2] + o, hz_n);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            st4_stream(a.d_w[c] + o, dn[c]);
            if (!MUR || skip[c] == 0u) {
                st4_stream(a.e_w[c] + o, en[c]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (!((skip[c] >> i) & 1u)) a.e_w[c][o + i] = en[c][i];
            }
        }
        // the H halo plane below a slab is kept current here, so H is never exchanged
        if (k == 0 && g.has_lower) {
            *(f4 *)(a.h_w[0] + o - ps) = hx_nz;
            *(f4 *)(a.h_w[1] + o - ps) = hy_nz;
        }
    }
}


} // namespace

void launch_sweep3d(const Fused3D &a, void *stream)
{
    if (a.k_end <= a.k_begin) return;
    const unsigned spaces = (unsigned)((a.k_end - a.k_begin + a.flat - 1) / a.flat);
    const dim3 grid((unsigned)a.seg_blocks * 8u * spaces + (a.ring ? 1u : 0u));
    const dim3 block((unsigned)a.block);
    static const int waves = std::getenv("FDTD_SWEEP3D_WAVES") ? std::atoi(std::getenv("FDTD_SWEEP3D_WAVES")) : 3;
#define GO(W, MUR, FLAT)                                                                                     \
    do {                                                                                                     \
        if (a.ntab == 1)                                                                                     \
            hipLaunchKernelGGL((fdtd3d_sweep<true, W, MUR, FLAT>), grid, block, 0, (hipStream_t)stream, a);  \
        else                                                                                                 \
            hipLaunchKernelGGL((fdtd3d_sweep<false, W, MUR, FLAT>), grid, block, 0, (hipStream_t)stream, a); \
    } while (0)
    // the Mur cases keep the neighbour words of the H update alive to the end: two waves per SIMD
    const bool flat = a.flat > 1;
    if (a.mur && flat)
        GO(2, true, true);
    else if (a.mur)
        GO(2, true, false);
    else if (flat)
        GO(3, false, true);
    else if (waves == 2)
        GO(2, false, false);
    else
        GO(3, false, false);
#undef GO
}

const char *sweep3d_kernel_name() { return "fdtd3d_sweep"; }

} // namespace fdtd
