// Feasibility probe: TWO time steps of the 3-D update (fdtd_calculations.cu:32-184: H, D, E with the
// reference's mixed fp32/fp64 Debye update, one material) in ONE pass over memory.
//
// A workgroup of 512 threads owns an x-y tile and marches in z.  A thread holds one 16-byte word
// (four cells in x) of the tile's plane: TX = 16 words x TY = 32 rows; one word on either side in x
// and two rows on either side in y go stale over the two steps, so a tile owns 14 x 28 words.
// Stage 1 (step n -> n+1) works on plane p, stage 2 (n+1 -> n+2) on plane p-1; what stage 2 needs
// of plane p-1 stays in registers from the iteration before.  x neighbours come from lane
// shuffles, the y+1 neighbours of the INPUT level from memory (cache hits: the thread above loads
// the same line), the y-1 / y+1 neighbours of COMPUTED values through LDS (two barriers per plane).
// A pass reads 15 words per cell (H, two levels of E and D) and writes 15 (H, levels n+1 and n+2).
//
// Guards of the reference at the grid faces, sources and Mur are NOT applied: results are compared
// with two passes of a plain one-step kernel pair (same arithmetic helpers) on the cells at least
// four cells away from every face, bit for bit.  Prints the time per pass of both and the
// cell-updates/s of each; the one-step pair here is a plain reference, not the product's sweep.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I fdtd_amd/csrc
//        tools/sweep3d_tb_probe.hip -o tools/sweep3d_tb_probe
// Run:   tools/sweep3d_tb_probe NX NY NZ [segment planes] [passes] [check: 1/0]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "device_math.h"

using fdtd::f4;
using fdtd::MatEntry;

#define CHK(x)                                                                                         \
    do {                                                                                               \
        hipError_t e_ = (x);                                                                           \
        if (e_ != hipSuccess) {                                                                        \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));                 \
            exit(1);                                                                                   \
        }                                                                                              \
    } while (0)

struct Grid {
    int nx, ny, nz, roww; // roww: 16-byte words per row
    float dt, kh_x, kh_y, kh_z, cx, cy, cz;
    MatEntry m;
};

// word index of (z, y, component, word in row): the three components of a field interleaved row by row
__host__ __device__ inline size_t widx(const Grid &g, int z, int y, int c, int xw)
{
    return (((size_t)z * g.ny + y) * 3 + c) * g.roww + xw;
}

struct Pass {
    Grid g;
    const f4 *h_r, *e0, *e1, *d0, *d1; // H, E^n, E^n-1, D^n, D^n-1
    f4 *h_w, *e_a, *e_b, *d_a, *d_b;   // H new, E^n+1, E^n+2, D^n+1, D^n+2
    int seg;                           // planes per z segment
    int tiles_x, tiles_y;
    int xcd_map;
};

// x neighbours of a word: cells x+1 (from the lane above) and x-1 (from the lane below)
__device__ __forceinline__ f4 xp(const f4 &v)
{
    f4 r;
    r.x = v.y, r.y = v.z, r.z = v.w, r.w = __shfl_down(v.x, 1);
    return r;
}
__device__ __forceinline__ f4 xm(const f4 &v)
{
    f4 r;
    r.y = v.x, r.z = v.y, r.w = v.z, r.x = __shfl_up(v.w, 1);
    return r;
}

// fdtd_calculations.cu:48-75 (left to right, fp32; both hx terms with dt/(mu0 dy) as written there)
__device__ __forceinline__ f4 upd_hx(const Grid &g, f4 hx, f4 ez, f4 ez_y1, f4 ey, f4 ey_z1)
{
    return (hx - g.kh_y * (ez_y1 - ez)) + g.kh_y * (ey_z1 - ey);
}
__device__ __forceinline__ f4 upd_hy(const Grid &g, f4 hy, f4 ex, f4 ex_z1, f4 ez, f4 ez_x1)
{
    return (hy - g.kh_z * (ex_z1 - ex)) + g.kh_x * (ez_x1 - ez);
}
__device__ __forceinline__ f4 upd_hz(const Grid &g, f4 hz, f4 ey, f4 ey_x1, f4 ex, f4 ex_y1)
{
    return (hz - g.kh_x * (ey_x1 - ey)) + g.kh_y * (ex_y1 - ex);
}
// fdtd_calculations.cu:92-120
__device__ __forceinline__ f4 upd_dx(const Grid &g, f4 dx, f4 hz, f4 hz_ym, f4 hy, f4 hy_zm)
{
    return (dx + g.cy * (hz - hz_ym)) - g.cz * (hy - hy_zm);
}
__device__ __forceinline__ f4 upd_dy(const Grid &g, f4 dy, f4 hx, f4 hx_zm, f4 hz, f4 hz_xm)
{
    return (dy + g.cz * (hx - hx_zm)) - g.cx * (hz - hz_xm);
}
__device__ __forceinline__ f4 upd_dz(const Grid &g, f4 dz, f4 hy, f4 hy_xm, f4 hx, f4 hx_ym)
{
    return (dz + g.cx * (hy - hy_xm)) - g.cy * (hx - hx_ym);
}
__device__ __forceinline__ f4 upd_e(const Grid &g, f4 e0, f4 e1, f4 dn, f4 d0, f4 d1)
{
    f4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = fdtd::e_update(g.m, g.dt, e0[i], e1[i], dn[i], d0[i], d1[i]);
    return r;
}

// ---- plain one-step reference: H kernel, then D + E kernel, one thread per word ----------------
__global__ void ref_h(Grid g, const f4 *h_r, f4 *h_w, const f4 *e0)
{
    const long n = (long)g.nz * g.ny * g.roww;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int xw = i % g.roww, y = (i / g.roww) % g.ny, z = i / ((long)g.roww * g.ny);
    auto ld = [&](const f4 *a, int zz, int yy, int c, int xx) -> f4 {
        const f4 zero = {0, 0, 0, 0};
        if (zz < 0 || zz >= g.nz || yy < 0 || yy >= g.ny || xx < 0 || xx >= g.roww) return zero;
        return a[widx(g, zz, yy, c, xx)];
    };
    auto x1 = [&](const f4 *a, int c) -> f4 {
        const f4 v = ld(a, z, y, c, xw), w = ld(a, z, y, c, xw + 1);
        f4 r = {v.y, v.z, v.w, w.x};
        return r;
    };
    const f4 ex = ld(e0, z, y, 0, xw), ey = ld(e0, z, y, 1, xw), ez = ld(e0, z, y, 2, xw);
    h_w[widx(g, z, y, 0, xw)] = upd_hx(g, ld(h_r, z, y, 0, xw), ez, ld(e0, z, y + 1, 2, xw), ey, ld(e0, z + 1, y, 1, xw));
    h_w[widx(g, z, y, 1, xw)] = upd_hy(g, ld(h_r, z, y, 1, xw), ex, ld(e0, z + 1, y, 0, xw), ez, x1(e0, 2));
    h_w[widx(g, z, y, 2, xw)] = upd_hz(g, ld(h_r, z, y, 2, xw), ey, x1(e0, 1), ex, ld(e0, z, y + 1, 0, xw));
}

__global__ void ref_de(Grid g, const f4 *h, const f4 *e0, const f4 *e1, const f4 *d0, const f4 *d1, f4 *e_w, f4 *d_w)
{
    const long n = (long)g.nz * g.ny * g.roww;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int xw = i % g.roww, y = (i / g.roww) % g.ny, z = i / ((long)g.roww * g.ny);
    auto ld = [&](const f4 *a, int zz, int yy, int c, int xx) -> f4 {
        const f4 zero = {0, 0, 0, 0};
        if (zz < 0 || zz >= g.nz || yy < 0 || yy >= g.ny || xx < 0 || xx >= g.roww) return zero;
        return a[widx(g, zz, yy, c, xx)];
    };
    auto xm1 = [&](const f4 *a, int c) -> f4 {
        const f4 v = ld(a, z, y, c, xw), w = ld(a, z, y, c, xw - 1);
        f4 r = {w.w, v.x, v.y, v.z};
        return r;
    };
    const f4 hx = ld(h, z, y, 0, xw), hy = ld(h, z, y, 1, xw), hz = ld(h, z, y, 2, xw);
    f4 dn[3];
    dn[0] = upd_dx(g, ld(d0, z, y, 0, xw), hz, ld(h, z, y - 1, 2, xw), hy, ld(h, z - 1, y, 1, xw));
    dn[1] = upd_dy(g, ld(d0, z, y, 1, xw), hx, ld(h, z - 1, y, 0, xw), hz, xm1(h, 2));
    dn[2] = upd_dz(g, ld(d0, z, y, 2, xw), hy, xm1(h, 1), hx, ld(h, z, y - 1, 0, xw));
    for (int c = 0; c < 3; ++c) {
        d_w[widx(g, z, y, c, xw)] = dn[c];
        e_w[widx(g, z, y, c, xw)] = upd_e(g, ld(e0, z, y, c, xw), ld(e1, z, y, c, xw), dn[c], ld(d0, z, y, c, xw), ld(d1, z, y, c, xw));
    }
}

// ---- the two-step sweep ---------------------------------------------------------------------------
constexpr int TX = 16, TY = 32, OWNX = TX - 2, OWNY = TY - 4;

struct V3 {
    f4 x, y, z;
};

// NT: stores with the nontemporal hint.  PREFETCH 1: the next plane's loads before stage 2 instead of after it
// (256 registers, spills).  PREFETCH 2: only what stage 1 needs first (H, the E plane above, the y+1 words)
// before stage 2, the rest at the top of the iteration, and E^n, D^n of the plane stage 2 works on parked in
// LDS (48 KB more) to make room.
template <bool NT, int PREFETCH> __global__ __launch_bounds__(TX *TY, 2) void sweep3d_two(const Pass a)
{
    const Grid &g = a.g;
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    __shared__ f4 ex1[2][TX * TY]; // stage 1: new hz, hx of the plane (for the thread above)
    __shared__ f4 ex2[2][TX * TY]; // E^n+1 ez, ex of the plane stage 1 did last (for the thread below)
    __shared__ f4 ex3[2][TX * TY]; // stage 2: new hz, hx
    __shared__ f4 parkq[PREFETCH == 2 ? 6 : 1][TX * TY]; // E^n, D^n of plane p-1 (own slots: no barrier)

    const int tid = threadIdx.x, tx = tid & (TX - 1), ty = tid / TX;
    // blocks: tile index fastest within a segment, so that neighbouring tiles run at the same time
    const int tiles = a.tiles_x * a.tiles_y;
    int tile, segi;
    if (a.xcd_map) {
        // block ids are dealt to the 8 XCDs in turn: XCD i walks the run i of the (padded) tile list of
        // a segment, so that tiles next to each other share an L2
        const int per = (tiles + 7) / 8, xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
        segi = q / per;
        tile = xcd * per + q % per;
        if (tile >= tiles) return;
    } else {
        tile = blockIdx.x % tiles, segi = blockIdx.x / tiles;
    }
    const int tix = tile % a.tiles_x, tiy = tile / a.tiles_x;
    const int xw = tix * OWNX - 1 + tx, y = tiy * OWNY - 2 + ty;
    const int z0 = segi * a.seg, z1 = min(z0 + a.seg, g.nz);
    const bool inxy = xw >= 0 && xw < g.roww && y >= 0 && y < g.ny;
    const bool own = inxy && tx >= 1 && tx <= OWNX && ty >= 2 && ty < 2 + OWNY;
    const bool up_ok = inxy && y + 1 < g.ny;
    const int up = tid + TX < TX * TY ? tid + TX : tid, dn = tid >= TX ? tid - TX : tid;

    auto ld = [&](const f4 *p, int z, int yy, int c, bool ok) -> f4 {
        if (!ok || z < 0 || z >= g.nz) return zero;
        const f4 *q = p + widx(g, z, yy, c, xw);
        return *q; // loads stay cached: the halo words of a tile are its neighbours' owned words
    };
    auto ld3 = [&](const f4 *p, int z) -> V3 {
        V3 v;
        v.x = ld(p, z, y, 0, inxy), v.y = ld(p, z, y, 1, inxy), v.z = ld(p, z, y, 2, inxy);
        return v;
    };

    // carried from iteration to iteration
    V3 e_cur = ld3(a.e0, z0 - 2);                    // E^n of the plane stage 1 works on
    V3 s_e1 = {zero, zero, zero}, s_h1 = s_e1, s_d1 = s_e1, s_en = s_e1, s_dn = s_e1; // plane p-1: E^n+1, H^n+1, D^n+1, E^n, D^n
    f4 hx2_zm = zero, hy2_zm = zero;                 // H^n+2 of plane p-2

    // what stage 1 loads for a plane
    struct In {
        V3 h_n, e_up, e_pr, d_n, d_pr;
        f4 ez_y1, ex_y1;
    };
    auto load_a = [&](In &in, int p) {
        in.h_n = ld3(a.h_r, p), in.e_up = ld3(a.e0, p + 1);
        in.ez_y1 = ld(a.e0, p, y + 1, 2, up_ok), in.ex_y1 = ld(a.e0, p, y + 1, 0, up_ok);
    };
    auto load_b = [&](In &in, int p) { in.e_pr = ld3(a.e1, p), in.d_n = ld3(a.d0, p), in.d_pr = ld3(a.d1, p); };
    auto load = [&](In &in, int p) {
        load_a(in, p);
        load_b(in, p);
    };
    In cur;
    if (PREFETCH == 2)
        load_a(cur, z0 - 2);
    else
        load(cur, z0 - 2);

    for (int p = z0 - 2; p <= z1; ++p) {
        if (PREFETCH == 2) load_b(cur, p);
        // ---- stage 1 on plane p ----
        V3 h1;
        h1.x = upd_hx(g, cur.h_n.x, e_cur.z, cur.ez_y1, e_cur.y, cur.e_up.y);
        h1.y = upd_hy(g, cur.h_n.y, e_cur.x, cur.e_up.x, e_cur.z, xp(e_cur.z));
        h1.z = upd_hz(g, cur.h_n.z, e_cur.y, xp(e_cur.y), e_cur.x, cur.ex_y1);
        ex1[0][tid] = h1.z;
        ex1[1][tid] = h1.x;
        // E^n+1 of plane p-1 at y+1 (written in the iteration before), taken out before it is replaced
        const f4 e1z_y1 = ex2[0][up], e1x_y1 = ex2[1][up];
        __syncthreads();
        const f4 hz1_ym = ex1[0][dn], hx1_ym = ex1[1][dn];
        V3 d1, e1;
        d1.x = upd_dx(g, cur.d_n.x, h1.z, hz1_ym, h1.y, s_h1.y);
        d1.y = upd_dy(g, cur.d_n.y, h1.x, s_h1.x, h1.z, xm(h1.z));
        d1.z = upd_dz(g, cur.d_n.z, h1.y, xm(h1.y), h1.x, hx1_ym);
        e1.x = upd_e(g, e_cur.x, cur.e_pr.x, d1.x, cur.d_n.x, cur.d_pr.x);
        e1.y = upd_e(g, e_cur.y, cur.e_pr.y, d1.y, cur.d_n.y, cur.d_pr.y);
        e1.z = upd_e(g, e_cur.z, cur.e_pr.z, d1.z, cur.d_n.z, cur.d_pr.z);
        ex2[0][tid] = e1.z;
        ex2[1][tid] = e1.x;
        const V3 e_n = e_cur, d_n = cur.d_n;
        e_cur = cur.e_up;
        // the next plane's loads go out now: stage 2 below runs while they are in flight
        if (PREFETCH == 1 && p < z1) load(cur, p + 1);
        if (PREFETCH == 2 && p < z1) load_a(cur, p + 1);

        // ---- stage 2 on plane p-1 ----
        V3 h2;
        h2.x = upd_hx(g, s_h1.x, s_e1.z, e1z_y1, s_e1.y, e1.y);
        h2.y = upd_hy(g, s_h1.y, s_e1.x, e1.x, s_e1.z, xp(s_e1.z));
        h2.z = upd_hz(g, s_h1.z, s_e1.y, xp(s_e1.y), s_e1.x, e1x_y1);
        ex3[0][tid] = h2.z;
        ex3[1][tid] = h2.x;
        __syncthreads();
        const f4 hz2_ym = ex3[0][dn], hx2_ym = ex3[1][dn];
        V3 d2, e2;
        if (PREFETCH == 2) {
            s_en.x = parkq[0][tid], s_en.y = parkq[1][tid], s_en.z = parkq[2][tid];
            s_dn.x = parkq[3][tid], s_dn.y = parkq[4][tid], s_dn.z = parkq[5][tid];
        }
        d2.x = upd_dx(g, s_d1.x, h2.z, hz2_ym, h2.y, hy2_zm);
        d2.y = upd_dy(g, s_d1.y, h2.x, hx2_zm, h2.z, xm(h2.z));
        d2.z = upd_dz(g, s_d1.z, h2.y, xm(h2.y), h2.x, hx2_ym);
        e2.x = upd_e(g, s_e1.x, s_en.x, d2.x, s_d1.x, s_dn.x);
        e2.y = upd_e(g, s_e1.y, s_en.y, d2.y, s_d1.y, s_dn.y);
        e2.z = upd_e(g, s_e1.z, s_en.z, d2.z, s_d1.z, s_dn.z);
        const int k = p - 1;
        if (own && k >= z0 && k < z1) {
            auto st = [&](f4 *q, int c, const f4 &v) {
                f4 *t = q + widx(g, k, y, c, xw);
                if (NT)
                    __builtin_nontemporal_store(v, t);
                else
                    *t = v;
            };
            st(a.h_w, 0, h2.x), st(a.h_w, 1, h2.y), st(a.h_w, 2, h2.z);
            st(a.e_b, 0, e2.x), st(a.e_b, 1, e2.y), st(a.e_b, 2, e2.z);
            st(a.e_a, 0, s_e1.x), st(a.e_a, 1, s_e1.y), st(a.e_a, 2, s_e1.z);
            st(a.d_b, 0, d2.x), st(a.d_b, 1, d2.y), st(a.d_b, 2, d2.z);
            st(a.d_a, 0, s_d1.x), st(a.d_a, 1, s_d1.y), st(a.d_a, 2, s_d1.z);
        }
        hx2_zm = h2.x, hy2_zm = h2.y;
        s_e1 = e1, s_h1 = h1, s_d1 = d1;
        if (PREFETCH == 2) {
            parkq[0][tid] = e_n.x, parkq[1][tid] = e_n.y, parkq[2][tid] = e_n.z;
            parkq[3][tid] = d_n.x, parkq[4][tid] = d_n.y, parkq[5][tid] = d_n.z;
        } else {
            s_en = e_n, s_dn = d_n;
        }
        if (PREFETCH == 0 && p < z1) load(cur, p + 1);
    }
}

// flat copy with the same stream mix: 15 arrays read, 15 written
__global__ void flat30(const f4 *const *src, f4 *const *dst, long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int k = 0; k < 5; ++k)
        for (int c = 0; c < 3; ++c) {
            const f4 v = __builtin_nontemporal_load(src[k] + 3 * i + c);
            __builtin_nontemporal_store(v, dst[k] + 3 * i + c);
        }
}

int main(int argc, char **argv)
{
    Grid g;
    g.nx = argc > 1 ? atoi(argv[1]) : 256;
    g.ny = argc > 2 ? atoi(argv[2]) : 256;
    g.nz = argc > 3 ? atoi(argv[3]) : 256;
    const int seg = argc > 4 ? atoi(argv[4]) : 64;
    const int passes = argc > 5 ? atoi(argv[5]) : 10;
    g.roww = (g.nx + 3) / 4;
    // the reference's constants for the shipped parameters (values only matter for realism)
    g.dt = 1.92450092e-12f;
    const float dx = 1e-3f, mu0 = 1.25663706e-6f;
    g.kh_x = g.kh_y = g.kh_z = g.dt / (mu0 * dx);
    g.cx = g.cy = g.cz = g.dt / dx;
    // a Debye tissue: a, c, d, tau as make_coef would give them (plausible magnitudes)
    g.m.a = 2.0f * 8.854e-12f * 4.0f * 7e-12f;
    const float b = g.m.a + 2.0f * g.dt * (8.854e-12f * 50.0f + 0.5f * 7e-12f);
    g.m.c = 1.0f / (b + 0.5f * g.dt * g.dt);
    g.m.d = g.m.a + b - 0.5f * g.dt * g.dt;
    g.m.tau = 7e-12f;

    const size_t words = (size_t)g.nz * g.ny * 3 * g.roww, bytes = words * sizeof(f4);
    printf("grid %d x %d x %d, %.2f GB per field (3 components), tile %d x %d words/rows owning %d x %d, segment %d planes\n", g.nx,
           g.ny, g.nz, bytes / 1e9, TX, TY, OWNX, OWNY, seg);
    // inputs: H, E0, E1, D0, D1; reference outputs after step 1 and 2; sweep outputs
    enum { H, E0, E1, D0, D1, RH1, RE1, RD1, RH2, RE2, RD2, SH, SEA, SEB, SDA, SDB, NA };
    f4 *d[NA];
    for (int i = 0; i < NA; ++i) CHK(hipMalloc(&d[i], bytes));
    {
        std::vector<float> host(words * 4);
        unsigned s = 12345u;
        const float scale[5] = {1e-3f, 1.0f, 1.0f, 1e-11f, 1e-11f};
        for (int i = 0; i < 5; ++i) {
            for (auto &v : host) {
                s = s * 1664525u + 1013904223u;
                v = scale[i] * ((int)(s >> 8) % 2001 - 1000) * 1e-3f;
            }
            CHK(hipMemcpy(d[i], host.data(), bytes, hipMemcpyHostToDevice));
        }
    }
    for (int i = 5; i < NA; ++i) CHK(hipMemset(d[i], 0, bytes));

    hipEvent_t t0, t1;
    CHK(hipEventCreate(&t0));
    CHK(hipEventCreate(&t1));
    const long nw = (long)g.nz * g.ny * g.roww;
    const dim3 rb(256), rg((unsigned)((nw + 255) / 256));
    auto ref_two = [&]() {
        hipLaunchKernelGGL(ref_h, rg, rb, 0, 0, g, d[H], d[RH1], d[E0]);
        hipLaunchKernelGGL(ref_de, rg, rb, 0, 0, g, d[RH1], d[E0], d[E1], d[D0], d[D1], d[RE1], d[RD1]);
        hipLaunchKernelGGL(ref_h, rg, rb, 0, 0, g, d[RH1], d[RH2], d[RE1]);
        hipLaunchKernelGGL(ref_de, rg, rb, 0, 0, g, d[RH2], d[RE1], d[E0], d[RD1], d[D0], d[RE2], d[RD2]);
    };
    ref_two();
    CHK(hipDeviceSynchronize());

    Pass a;
    a.g = g;
    a.h_r = d[H], a.e0 = d[E0], a.e1 = d[E1], a.d0 = d[D0], a.d1 = d[D1];
    a.h_w = d[SH], a.e_a = d[SEA], a.e_b = d[SEB], a.d_a = d[SDA], a.d_b = d[SDB];
    a.seg = seg;
    a.tiles_x = (g.roww + OWNX - 1) / OWNX;
    a.tiles_y = (g.ny + OWNY - 1) / OWNY;
    const int segs = (g.nz + seg - 1) / seg;
    a.xcd_map = getenv("PROBE_XCD") && atoi(getenv("PROBE_XCD"));
    const dim3 sg((unsigned)(a.xcd_map ? 8 * ((a.tiles_x * a.tiles_y + 7) / 8) * segs : a.tiles_x * a.tiles_y * segs)), sb(TX * TY);
    printf("two-step sweep: %d x %d tiles x %d segments = %u workgroups of %d threads; owned share of the word slots %.3f\n", a.tiles_x,
           a.tiles_y, segs, sg.x, TX * TY, (double)g.roww * g.ny / ((double)a.tiles_x * a.tiles_y * TX * TY));
    const bool nt = getenv("PROBE_NT") && atoi(getenv("PROBE_NT"));
    const int pf = getenv("PROBE_PREFETCH") ? atoi(getenv("PROBE_PREFETCH")) : 0;
    auto sweep = [&]() {
        if (nt && pf == 2)
            hipLaunchKernelGGL((sweep3d_two<true, 2>), sg, sb, 0, 0, a);
        else if (nt)
            hipLaunchKernelGGL((sweep3d_two<true, 0>), sg, sb, 0, 0, a);
        else if (pf == 2)
            hipLaunchKernelGGL((sweep3d_two<false, 2>), sg, sb, 0, 0, a);
        else if (pf == 1)
            hipLaunchKernelGGL((sweep3d_two<false, 1>), sg, sb, 0, 0, a);
        else
            hipLaunchKernelGGL((sweep3d_two<false, 0>), sg, sb, 0, 0, a);
    };
    sweep();
    CHK(hipDeviceSynchronize());

    // compare on cells at least four cells from every face
    if (!(argc > 6 && atoi(argv[6]) == 0)) {
        std::vector<float> r(words * 4), s(words * 4);
        const int pairs[5][2] = {{RH2, SH}, {RE1, SEA}, {RE2, SEB}, {RD1, SDA}, {RD2, SDB}};
        const char *names[5] = {"H n+2", "E n+1", "E n+2", "D n+1", "D n+2"};
        for (int q = 0; q < 5; ++q) {
            CHK(hipMemcpy(r.data(), d[pairs[q][0]], bytes, hipMemcpyDeviceToHost));
            CHK(hipMemcpy(s.data(), d[pairs[q][1]], bytes, hipMemcpyDeviceToHost));
            long bad = 0, seen = 0, nonzero = 0;
            for (int z = 4; z < g.nz - 4; ++z)
                for (int y = 4; y < g.ny - 4; ++y)
                    for (int c = 0; c < 3; ++c)
                        for (int x = 4; x < g.nx - 4; ++x) {
                            const size_t i = widx(g, z, y, c, x / 4) * 4 + (x & 3);
                            ++seen;
                            nonzero += r[i] != 0.0f;
                            if (memcmp(&r[i], &s[i], 4)) {
                                if (bad < 5) printf("  %s differs at (%d,%d,%d) comp %d: ref %.9g sweep %.9g\n", names[q], x, y, z, c, r[i], s[i]);
                                ++bad;
                            }
                        }
            printf("%s: %ld of %ld interior values differ (%ld non-zero)\n", names[q], bad, seen, nonzero);
        }
    }

    auto time_of = [&](auto fn, int n) {
        fn();
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(t0));
        for (int i = 0; i < n; ++i) fn();
        CHK(hipEventRecord(t1));
        CHK(hipEventSynchronize(t1));
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, t0, t1));
        return ms / n;
    };
    const double cells = (double)g.nx * g.ny * g.nz;
    const float ms_ref = time_of(ref_two, passes), ms_sw = time_of(sweep, passes);
    printf("plain one-step kernels, two steps: %.3f ms (%.1f Gcell-updates/s)\n", ms_ref, 2 * cells / ms_ref / 1e6);
    printf("two-step sweep, one pass:          %.3f ms (%.1f Gcell-updates/s; 121 B x cells / time = %.0f GB/s)\n", ms_sw,
           2 * cells / ms_sw / 1e6, cells * 121.0 / ms_sw / 1e6);
    {
        const f4 *srcs[5] = {d[H], d[E0], d[E1], d[D0], d[D1]};
        f4 *dsts[5] = {d[SH], d[SEA], d[SEB], d[SDA], d[SDB]};
        const f4 **ds;
        f4 **dd;
        CHK(hipMalloc(&ds, sizeof(srcs)));
        CHK(hipMalloc(&dd, sizeof(dsts)));
        CHK(hipMemcpy(ds, srcs, sizeof(srcs), hipMemcpyHostToDevice));
        CHK(hipMemcpy(dd, dsts, sizeof(dsts), hipMemcpyHostToDevice));
        const long n = (long)words / 3;
        auto flat = [&]() { hipLaunchKernelGGL(flat30, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, ds, dd, n); };
        const float ms_flat = time_of(flat, passes);
        printf("flat copy of the same 15 + 15 arrays: %.3f ms (%.0f GB/s)\n", ms_flat, 2.0 * 5 * bytes / ms_flat / 1e6);
    }
    return 0;
}
