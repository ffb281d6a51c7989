// fdtd2d_sweep: one pass over memory per time step of the 2-D TMz grid --
// updateHField + updateDField + updateSources + updateEField
// (fdtd_calculations.cu:32-213) fused.
//
// Work decomposition.  A wave owns 256 consecutive columns of one row (lane l
// holds columns 4l..4l+3, one 16-byte word per array) and `chunk_rows` rows,
// one after the other; a workgroup is four such waves side by side.  Blocks are
// numbered so that at any moment the blocks in flight cover a compact band of
// rows of every array -- the access pattern of a flat copy, which is what HBM
// (and the TLBs) serve fastest; with chunk_rows = 1 the band is as narrow as it
// can be.  Inside a band of 32 rows the hardware's round-robin of block ids
// over the 8 XCDs is undone, so that vertically adjacent blocks share an L2.
//
// Neighbours.  dz^{n+1}(x, y) needs the NEW hy at x-1 and the NEW hx at y-1.
// x: from the adjacent lane (shuffle); the lane at the left edge of the strip
// recomputes hy(x-1) from hy, ez^n left of the strip, the lane at the right edge
// fetches ez^n right of it.  y: the wave recomputes the new hx of the row below
// its first row from hx and ez^n of that row (served by the L2 the block below
// just filled), and carries hx and ez^n in registers between its own rows.  H
// is read from one generation and written to the other, so a recomputed value
// never races with the owner's store.
//
// Traffic per cell: hx, hy, ez^n, ez^{n-1}, dz^n, dz^{n-1} in (24 B) + 1 B
// material code, hx, hy, ez^{n+1}, dz^{n+1} out (16 B) = 41 B from HBM, plus
// the re-read of ez^n (two rows) and hx (one row) from cache.
//
// Materials.  A wave whose 256 cells hold one material code takes that
// material's four coefficients through scalar loads; a mixed wave gathers one
// 16-byte table row per cell (the table stays in L1/L2).
#include <cstdlib>

#include "device_math.h"

namespace fdtd {

namespace {

// Everything a wave loads for one row.
struct Row2D {
    f4 ezn;             // ez^n of the row above
    f4 hxo, hyo;        // H, read generation
    f4 e1, d0, d1;      // ez^{n-1}, dz^n, dz^{n-1}
    unsigned cd;        // four material codes
    float e_edge;       // ez^n right of the strip (lane 63)
    float hy_lo, ez_lo; // hy and ez^n left of the strip (lane 0)
};

// POL selects which touch of the re-read arrays carries the nontemporal hint:
// 0 = the first (the streaming loads of ez^n and hx), 1 = the last (the loads of
// the row below, after which no block reads those lines again).
template <int POL>
__device__ __forceinline__ void load_row(Row2D &r, const Sweep2D &a, size_t off, size_t coff, size_t pitch, bool act,
                                         bool edge_r, bool edge_l)
{
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    if (POL == 0) {
        r.ezn = act ? ld4_stream(a.ez_n + off + pitch) : zero;
        r.hxo = act ? ld4_stream(a.hx_r + off) : zero;
    } else {
        r.ezn = act ? *(const f4 *)(a.ez_n + off + pitch) : zero;
        r.hxo = act ? *(const f4 *)(a.hx_r + off) : zero;
    }
    r.hyo = act ? ld4_stream(a.hy_r + off) : zero;
    r.e1 = act ? ld4_stream(a.ez_w + off) : zero;
    r.d0 = act ? ld4_stream(a.dz_n + off) : zero;
    r.d1 = act ? ld4_stream(a.dz_w + off) : zero;
    r.cd = act ? *(const unsigned *)(a.code + coff) : 0u;
    r.e_edge = edge_r ? a.ez_n[off + 4] : 0.f;
    r.hy_lo = edge_l ? a.hy_r[off - 1] : 0.f;
    r.ez_lo = edge_l ? a.ez_n[off - 1] : 0.f;
}

// How block ids map to (x block, row chunk).
struct BlockMap {
    int xblocks; // x blocks of 1024 columns that hold cells
    int xpad;    // row-major order: x blocks per row of blocks, padded
    int band;    // 0 = row-major order; else rows of blocks per band (a multiple of 8)
    int chunks;  // row chunks in all
};

// ONE: the whole grid is one material; its coefficients are a kernel argument.
template <bool ONE, int POL, int W>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W, W))) void fdtd2d_sweep(const Sweep2D a, const BlockMap m)
{
    const Grid2D &g = a.g;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    int xb, chunk;
    if (m.band == 0) {
        xb = (int)(blockIdx.x % (unsigned)m.xpad);
        chunk = (int)(blockIdx.x / (unsigned)m.xpad);
        if (xb >= m.xblocks) return;
    } else {
        // Bands of `band` rows.  Inside a band the blocks are ordered column by column and
        // cut into 8 equal runs; the hardware deals consecutive block ids to the 8 XCDs in
        // turn, so XCD i walks run i: consecutive rows of one column, one after the other.
        const unsigned per_band = (unsigned)m.band * (unsigned)m.xblocks;
        const unsigned s = blockIdx.x / per_band, bl = blockIdx.x % per_band;
        const unsigned q = (bl & 7u) * (per_band >> 3) + (bl >> 3);
        xb = (int)(q / (unsigned)m.band);
        chunk = (int)(s * (unsigned)m.band + q % (unsigned)m.band);
        if (chunk >= m.chunks) return;
    }
    const int xw0 = (xb * 4 + wave) * 256;
    if (xw0 >= g.nx) return;
    const int x4 = xw0 + lane * 4;
    const bool act = x4 < g.nx;
    const int j0 = a.row_begin + chunk * a.chunk_rows;
    const int j1 = min(j0 + a.chunk_rows, a.row_end);
    const size_t pitch = (size_t)g.pitch;

    // x guards of the four columns held by this lane
    bool m_hx[4], m_hy[4], m_ez[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int x = x4 + i;
        m_hx[i] = x >= 1 && x <= g.nx - 2; // also the dz guard
        m_hy[i] = x <= g.nx - 2;
        m_ez[i] = x >= 1 && x <= g.nx - 1;
    }
    const bool edge_r = (lane == 63) && (xw0 + 256 < g.nx);
    const bool edge_l = (lane == 0) && (xw0 > 0);

    // sources that fall into this wave's strip and rows (wave-uniform mask)
    unsigned srcmask = 0;
    for (int s = 0; s < a.src.count; ++s) {
        const int js = a.src.y[s] - g.gy0;
        if (a.src.x[s] >= xw0 && a.src.x[s] < xw0 + 256 && js >= j0 && js < j1) srcmask |= 1u << s;
    }

    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    size_t off = (size_t)(j0 + 1) * pitch + x4;
    size_t coff = (size_t)(j0 + 1) * g.cpitch + x4;

    // Loads of the first row and of the row below it, all issued before any use.
    Row2D cur;
    load_row<POL>(cur, a, off, coff, pitch, act, edge_r, edge_l);
    f4 ezc = act ? *(const f4 *)(a.ez_n + off) : zero;
    f4 hxp = zero, ezm = zero;
    if (act) {
        hxp = POL ? ld4_stream(a.hx_r + off - pitch) : *(const f4 *)(a.hx_r + off - pitch);
        ezm = POL ? ld4_stream(a.ez_n + off - pitch) : *(const f4 *)(a.ez_n + off - pitch);
    }
    {
        // new hx of row j0 - 1 (owned by the block below, or the halo row)
        const int y = g.gy0 + j0 - 1;
        if (y >= 0 && y <= g.gny - 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (m_hx[i]) hxp[i] = hxp[i] - g.kh_y * (ezc[i] - ezm[i]);
        }
        if (j0 == 0 && g.has_lower && act) *(f4 *)(a.hx_w + off - pitch) = hxp;
    }

    // One row: H, then D (+ source), then E, all from registers.
    auto row = [&](const Row2D &r, int j, size_t o) {
        const int y = g.gy0 + j;
        float e_right = __shfl_down(ezc.x, 1);
        if (lane == 63) e_right = r.e_edge;

        const bool gy_hx = y <= g.gny - 2;
        const bool gy_in = y >= 1 && y <= g.gny - 2;

        f4 hxn, hyn;
        const float er[4] = {ezc.y, ezc.z, ezc.w, e_right};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            hxn[i] = (gy_hx && m_hx[i]) ? r.hxo[i] - g.kh_y * (r.ezn[i] - ezc[i]) : r.hxo[i];
            hyn[i] = (gy_in && m_hy[i]) ? r.hyo[i] + g.kh_x * (er[i] - ezc[i]) : r.hyo[i];
        }
        float hy_left = __shfl_up(hyn.w, 1);
        if (lane == 0) hy_left = (edge_l && gy_in) ? r.hy_lo + g.kh_x * (ezc.x - r.ez_lo) : r.hy_lo;
        const float hl[4] = {hy_left, hyn.x, hyn.y, hyn.z};

        unsigned srcflag = 0;
        if (srcmask) {
            for (int s = 0; s < a.src.count; ++s) {
                if (!((srcmask >> s) & 1u) || a.src.y[s] - g.gy0 != j) continue;
                const int rel = a.src.x[s] - xw0;
                if ((rel >> 2) == lane) srcflag |= 1u << (rel & 3);
            }
        }

        f4 dzn;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v = (r.d0[i] + g.cx * (hyn[i] - hl[i])) - g.cy * (hxn[i] - hxp[i]);
            float dn = (gy_in && m_hx[i]) ? v : r.d1[i];
            if ((srcflag >> i) & 1u) dn = v - a.src.jz;
            dzn[i] = dn;
        }

        f4 ezw;
        bool uniform = true;
        MatEntry mu = a.one;
        if (!ONE) {
            const unsigned c0 = (unsigned)__builtin_amdgcn_readfirstlane((int)r.cd) & 0xffu;
            uniform = __all(r.cd == c0 * 0x01010101u);
            if (uniform) mu = a.table[c0];
        }
        if (uniform) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                ezw[i] = m_ez[i] ? e_update(mu, g.dt, ezc[i], r.e1[i], dzn[i], r.d0[i], r.d1[i]) : r.e1[i];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const MatEntry m = a.table[(r.cd >> (8 * i)) & 0xffu];
                ezw[i] = m_ez[i] ? e_update(m, g.dt, ezc[i], r.e1[i], dzn[i], r.d0[i], r.d1[i]) : r.e1[i];
            }
        }

        if (act) {
            st4_stream(a.hx_w + o, hxn);
            st4_stream(a.hy_w + o, hyn);
            st4_stream(a.dz_w + o, dzn);
            st4_stream(a.ez_w + o, ezw);
        }
        hxp = hxn;
        ezc = r.ezn;
    };

    for (int j = j0; j < j1 - 1; ++j, off += pitch, coff += g.cpitch) {
        Row2D nxt;
        load_row<POL>(nxt, a, off + pitch, coff + g.cpitch, pitch, act, edge_r, edge_l);
        row(cur, j, off);
        cur = nxt;
    }
    row(cur, j1 - 1, off);
}

} // namespace

void launch_sweep2d(const Sweep2D &a, void *stream)
{
    BlockMap xblocks;
    if (a.row_end <= a.row_begin) return;
    xblocks.chunks = (a.row_end - a.row_begin + a.chunk_rows - 1) / a.chunk_rows;
    xblocks.xblocks = (a.g.nx + 1023) / 1024;
    xblocks.xpad = a.nopad ? xblocks.xblocks : (xblocks.xblocks + 7) / 8 * 8;
    xblocks.band = a.band;
    unsigned nblocks;
    if (a.band == 0)
        nblocks = (unsigned)xblocks.xpad * (unsigned)xblocks.chunks;
    else
        nblocks = (unsigned)((xblocks.chunks + a.band - 1) / a.band) * (unsigned)a.band * (unsigned)xblocks.xblocks;
    const dim3 grid(nblocks);
    const bool one = a.ntab == 1;
    // Waves per SIMD are pinned: more waves widen the band of rows in flight and DRAM serves a
    // compact band fastest (16384^2: 2 waves 1.885 ms, 3 1.794, 4 1.829, 5 1.925 per sweep).
    static const int waves = std::getenv("FDTD_SWEEP2D_WAVES") ? std::atoi(std::getenv("FDTD_SWEEP2D_WAVES")) : 3;
#define GO(POL, W)                                                                                          \
    do {                                                                                                    \
        if (one)                                                                                            \
            hipLaunchKernelGGL((fdtd2d_sweep<true, POL, W>), grid, dim3(256), 0, (hipStream_t)stream, a, xblocks);  \
        else                                                                                                \
            hipLaunchKernelGGL((fdtd2d_sweep<false, POL, W>), grid, dim3(256), 0, (hipStream_t)stream, a, xblocks); \
    } while (0)
    if (!(a.variant & 1))
        GO(0, 3);
    else if (waves == 5)
        GO(1, 5);
    else if (waves == 4)
        GO(1, 4);
    else if (waves == 2)
        GO(1, 2);
    else
        GO(1, 3);
#undef GO
}

const char *sweep2d_kernel_name() { return "fdtd2d_sweep"; }

} // namespace fdtd
