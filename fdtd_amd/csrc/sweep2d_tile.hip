// fdtd2d_tile: K time steps of a 2-D TMz grid in ONE launch, a tile per workgroup.
//
// Grids of a few million cells and less are not bound by HBM bandwidth but by
// latency: a one-step sweep, the Mur kernel and the probe kernel are three
// launches of a few microseconds each per step, and the row-by-row march of
// fdtd2d_sweep_tb waits for memory once per row.  Here a workgroup loads a tile
// of WX 16-byte words x TY rows once (every load of the tile is in flight at
// the same time), advances it K steps with the state of each word in the
// registers of one thread -- stage s applies updateHField + updateDField +
// updateSources + updateEField (fdtd_calculations.cu:32-213) and the Mur update
// (fdtd_calculations.cu:369-427) of step n+s, the same sequence of IEEE
// operations as the other 2-D kernels -- and stores the words it owns: levels
// n+K, n+K-1 and H.  One launch per pass: the Mur update of the last level and
// the probe rows of every level are written here too.
//
// Validity.  One ring of cells around the tile goes stale per step (their
// neighbours outside the tile are not advanced), so a tile owns the words that
// lie two words (8 columns) inside its left and right edges and K rows inside
// its lower and upper edges: K <= 8.  Cells beyond a face of the grid never
// change, so nothing goes stale from there.  The last tile of a row or column
// of tiles is anchored at the end of the range: it may own cells its neighbour
// owns as well; both compute and store the same values.
//
// Neighbours.  x: lane shuffles (a wave holds whole tile rows).  y: LDS -- ez of
// the current and of the next level in two buffers (the y-face Mur update
// needs both levels of the row next to the face), the new hx in a third; two
// barriers per step, three in tiles that hold a y face of the grid.
//
// Storage and arguments are those of the multi-step sweep (SweepTB): the pass
// reads one pair of E/D arrays and one H generation and writes the other.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "device_math.h"

namespace fdtd {

namespace {

template <int WX, int RT, int V, bool ONE>
__global__ __launch_bounds__(WX *RT) void fdtd2d_tile(const SweepTile t)
{
    constexpr int TY = RT * V, NT = WX * RT;
    static_assert(64 % WX == 0, "a wave holds whole tile rows");
    const SweepTB &a = t.b;
    const Grid2D &g = a.g;
    // rows 0 and TY + 1 of the buffers stand for the rows below and above the tile: what is read
    // from them only reaches cells that have gone stale anyway
    __shared__ f4 ezs[2][TY + 2][WX];
    __shared__ f4 hxs[TY + 2][WX];
    __shared__ MatEntry tab[ONE ? 1 : FDTD_MAX_CODES];
    // the field-independent factors of the Mur update (two square roots and a division per face cell)
    // are taken once per launch: [side][tile row] for the x faces, [side][column of the tile] for the y faces
    __shared__ f4 mur_xc[2][TY], mur_yc[2][WX * 4];
    const int tid = threadIdx.x;
    if (!ONE)
        for (int i = tid; i < FDTD_MAX_CODES; i += NT) tab[i] = a.table[i];

    const int K = t.steps;
    const int wx = tid % WX, trow = tid / WX;
    const int tx = blockIdx.x % t.tiles_x, ty = blockIdx.x / t.tiles_x;
    constexpr int OWNW = WX - 2 * kTbHaloWords;
    const int nw = (g.nx + 3) >> 2;
    const int own_w0 = min(tx * OWNW, max(0, nw - OWNW));
    const int own_j0 = min(a.y_begin + ty * t.own_rows, max(a.y_begin, a.y_end - t.own_rows));
    const int own_j1 = min(own_j0 + t.own_rows, a.y_end);
    const int w = own_w0 - kTbHaloWords + wx; // word of the row held by this thread
    const int x4 = w * 4;
    const int j0 = own_j0 - K;                // local row of tile row 0
    const bool own_x = wx >= kTbHaloWords && wx < kTbHaloWords + OWNW && w < nw;
    const long pitch = g.pitch;

    bool m_hx[4], m_hy[4], m_ez[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int x = x4 + i;
        m_hx[i] = x >= 1 && x <= g.nx - 2; // also the dz guard and the x range of the y-face Mur cells
        m_hy[i] = x >= 0 && x <= g.nx - 2;
        m_ez[i] = x >= 1 && x <= g.nx - 1;
    }
    // the x = nx-1 Mur cell: word and component, and whether this tile holds it
    const int xe_w = (g.nx - 1) >> 2, xe_c = (g.nx - 1) & 3;
    const int w_first = own_w0 - kTbHaloWords;
    const bool tile_xe = xe_w >= w_first && xe_w < w_first + WX;
    // the y faces of the grid inside this tile
    const int y_first = g.gy0 + j0;
    const bool tile_y0 = y_first <= 0 && 0 < y_first + TY;
    const bool tile_ye = y_first <= g.gny - 1 && g.gny - 1 < y_first + TY;

    // per thread: V words, one above the other RT rows apart
    f4 ez[V], ezp[V], dz[V], dzp[V], hx[V], hy[V];
    unsigned cd[V], srcbits[V], prb[V];
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int r = v * RT + trow, j = j0 + r;
        const bool act = w >= 0 && w < nw && j >= -t.halo_rows && j < g.nyl + t.halo_rows;
        const long off = (long)(j + 1) * pitch + x4;
        hx[v] = act ? *(const f4 *)(a.hx_r + off) : zero;
        hy[v] = act ? *(const f4 *)(a.hy_r + off) : zero;
        ez[v] = act ? *(const f4 *)(a.ez_n + off) : zero;
        ezp[v] = act ? *(const f4 *)(a.ez_p + off) : zero;
        dz[v] = act ? *(const f4 *)(a.dz_n + off) : zero;
        dzp[v] = act ? *(const f4 *)(a.dz_p + off) : zero;
        cd[v] = (!ONE && act) ? *(const unsigned *)(a.code + (long)(j + 1) * g.cpitch + x4) : 0u;
        srcbits[v] = 0;
        for (int s = 0; s < a.src.count; ++s)
            if (a.src.y[s] - g.gy0 == j && (a.src.x[s] >> 2) == w) srcbits[v] |= 1u << (a.src.x[s] & 3);
        prb[v] = 0;
        if (a.ring && own_x && j >= own_j0 && j < own_j1)
            for (int s = 0; s < a.probes.count; ++s)
                if (a.probes.y[s] - g.gy0 == j && a.probes.x[s] >= 0 && a.probes.x[s] < g.nx && (a.probes.x[s] >> 2) == w)
                    prb[v] |= 1u << s;
        ezs[0][r + 1][wx] = ez[v];
        const int y = g.gy0 + j;
        if (y >= 1 && y <= g.gny - 2) {
            if (w == 0) mur_xc[0][r] = mur_coefs(g.dt, a.dx, a.mu0, a.eps0, a.rpx0[(size_t)y * 2], a.rpx0[(size_t)y * 2 + 1]);
            if (w == xe_w) mur_xc[1][r] = mur_coefs(g.dt, a.dx, a.mu0, a.eps0, a.rpxe[(size_t)y * 2 + 1], a.rpxe[(size_t)y * 2]);
        }
        if (y == 0 || y == g.gny - 1) {
            const bool high = y != 0;
            const float *rp = high ? a.rpye : a.rpy0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (m_hx[i]) {
                    const int x = x4 + i;
                    mur_yc[high][wx * 4 + i] = mur_coefs(g.dt, a.dy, a.mu0, a.eps0, high ? rp[g.nx + x] : rp[x], high ? rp[x] : rp[g.nx + x]);
                }
        }
    }
    if (trow == 0) {
        ezs[0][0][wx] = ezs[1][0][wx] = hxs[0][wx] = zero;
        ezs[0][TY + 1][wx] = ezs[1][TY + 1][wx] = hxs[TY + 1][wx] = zero;
    }
    // probes that lie on another slab (or outside the grid): their rows are zeros
    if (a.ring && blockIdx.x == 0 && tid < a.probes.count) {
        const int j = a.probes.y[tid] - g.gy0;
        if (!(j >= 0 && j < g.nyl && a.probes.x[tid] >= 0 && a.probes.x[tid] < g.nx))
            for (int s = 1; s <= K; ++s) {
                float *dst = a.ring + ((size_t)(s < K ? a.slot[s] : t.slot_last) * FDTD_MAX_PROBES + tid) * 4;
                dst[0] = dst[1] = dst[2] = dst[3] = 0.f;
            }
    }
    __syncthreads();

    // `inner` (a compile-time flag): every cell of the tile lies inside the guards of every update,
    // away from the Mur faces, and the tile holds no source and no probe -- the guards fold away.
    auto advance = [&](auto inner) {
    constexpr bool IN = decltype(inner)::value;
    int cb = 0; // buffer holding ez of the level the step starts from
    for (int s = 1; s <= K; ++s) {
        const float jz = a.jz[s - 1];
        const int nb = cb ^ 1;
        f4 hyn[V];
        // H of step n+s: hx needs ez of the row above, hy ez of the column to the right
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int r = v * RT + trow, y = y_first + r;
            const bool gy_hx = y >= 0 && y <= g.gny - 2;
            const bool gy_in = y >= 1 && y <= g.gny - 2;
            const f4 ezu = ezs[cb][r + 2][wx];
            const float e_right = __shfl_down(ez[v].x, 1);
            const float er[4] = {ez[v].y, ez[v].z, ez[v].w, e_right};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                hx[v][i] = (IN || (gy_hx && m_hx[i])) ? hx[v][i] - g.kh_y * (ezu[i] - ez[v][i]) : hx[v][i];
                hyn[v][i] = (IN || (gy_in && m_hy[i])) ? hy[v][i] + g.kh_x * (er[i] - ez[v][i]) : hy[v][i];
            }
            hy[v] = hyn[v];
            hxs[r + 1][wx] = hx[v];
        }
        __syncthreads();
        // D, source, E of step n+s, Mur on the x faces
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int r = v * RT + trow, y = y_first + r;
            const bool gy_in = y >= 1 && y <= g.gny - 2;
            const bool gy_ez = y >= 0 && y <= g.gny - 1;
            const f4 hxb = hxs[r][wx];
            const float hy_left = __shfl_up(hy[v].w, 1);
            const float hl[4] = {hy_left, hy[v].x, hy[v].y, hy[v].z};
            MatEntry m[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) m[i] = ONE ? a.one : tab[(cd[v] >> (8 * i)) & 0xffu];
            f4 dn, en;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float val = (dz[v][i] + g.cx * (hy[v][i] - hl[i])) - g.cy * (hx[v][i] - hxb[i]);
                float d = (IN || (gy_in && m_hx[i])) ? val : dzp[v][i];
                if (!IN && ((srcbits[v] >> i) & 1u)) d = val - jz;
                dn[i] = d;
                en[i] = (IN || (gy_ez && m_ez[i])) ? e_update(m[i], g.dt, ez[v][i], ezp[v][i], d, dz[v][i], dzp[v][i]) : ezp[v][i];
            }
            // fdtd_calculations.cu:369-395
            const bool yok = y >= 1 && y <= g.gny - 2;
            if (!IN && w == 0 && yok) en.x = mur_apply(mur_xc[0][r], en.y, ez[v].y, ez[v].x);
            if (!IN && tile_xe) {
                const float t_prev = __shfl_up(en.w, 1), s_prev = __shfl_up(ez[v].w, 1);
                if (w == xe_w && yok) {
                    const f4 c = mur_xc[1][r];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (i == xe_c) {
                            const float t_in = i ? en[i ? i - 1 : 0] : t_prev, s_in = i ? ez[v][i ? i - 1 : 0] : s_prev;
                            en[i] = mur_apply(c, t_in, s_in, ez[v][i]);
                        }
                }
            }
            dzp[v] = dz[v];
            dz[v] = dn;
            ezp[v] = ez[v];
            ez[v] = en;
            ezs[nb][r + 1][wx] = en;
        }
        __syncthreads();
        // Mur on the y faces, fdtd_calculations.cu:401-427: the face row takes the new and the old
        // level of the row next to it from the two buffers
        if (!IN && (tile_y0 || tile_ye)) {
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const int r = v * RT + trow, y = y_first + r;
                const bool low = y == 0, high = y == g.gny - 1;
                if (!(low || high) || g.gny < 2) continue;
                const int rin = low ? r + 1 : r - 1;
                const f4 et_in = ezs[nb][rin + 1][wx], es_in = ezs[cb][rin + 1][wx];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (!m_hx[i]) continue;
                    ez[v][i] = mur_apply(mur_yc[high][wx * 4 + i], et_in[i], es_in[i], ezp[v][i]);
                }
                ezs[nb][r + 1][wx] = ez[v];
            }
            __syncthreads();
        }
        // probe rows of step n+s
#pragma unroll
        for (int v = 0; v < V; ++v) {
            if (IN || !prb[v]) continue;
            const int slot = s < K ? a.slot[s] : t.slot_last;
            for (int p = 0; p < a.probes.count; ++p) {
                if (!((prb[v] >> p) & 1u)) continue;
                const int i = a.probes.x[p] & 3;
                float *dst = a.ring + ((size_t)slot * FDTD_MAX_PROBES + p) * 4;
                float phx = hx[v].x, phy = hy[v].x, pdz = dz[v].x, pez = ez[v].x;
#pragma unroll
                for (int q = 1; q < 4; ++q)
                    if (q == i) phx = hx[v][q], phy = hy[v][q], pdz = dz[v][q], pez = ez[v][q];
                dst[0] = phx;
                dst[1] = phy;
                dst[2] = pdz;
                dst[3] = pez;
            }
        }
        cb = nb;
    }
    };
    bool any_src = false;
    for (int s = 0; s < a.src.count; ++s) {
        const int js = a.src.y[s] - g.gy0 - j0, ws = (a.src.x[s] >> 2) - w_first;
        any_src |= js >= 0 && js < TY && ws >= 0 && ws < WX;
    }
    if (a.ring)
        for (int s = 0; s < a.probes.count; ++s) {
            const int js = a.probes.y[s] - g.gy0 - j0, ws = (a.probes.x[s] >> 2) - w_first;
            any_src |= js >= 0 && js < TY && ws >= 0 && ws < WX;
        }
    const bool inner_tile = w_first >= 1 && (w_first + WX - 1) * 4 + 3 <= g.nx - 2 && y_first >= 1 &&
                            y_first + TY - 1 <= g.gny - 2 && !any_src;
    if (inner_tile)
        advance(std::true_type{});
    else
        advance(std::false_type{});

#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int j = j0 + v * RT + trow;
        if (!(own_x && j >= own_j0 && j < own_j1)) continue;
        const long q = (long)(j + 1) * pitch + x4;
        *(f4 *)(a.hx_w + q) = hx[v];
        *(f4 *)(a.hy_w + q) = hy[v];
        *(f4 *)(a.dz_b + q) = dz[v];
        *(f4 *)(a.ez_b + q) = ez[v];
        *(f4 *)(a.dz_a + q) = dzp[v];
        *(f4 *)(a.ez_a + q) = ezp[v];
    }
}

template <int WX, int RT, int V> int launch_tile(SweepTile t, hipStream_t stream)
{
    constexpr int TY = RT * V;
    t.own_rows = TY - 2 * t.steps;
    const int nw = (t.b.g.nx + 3) / 4, n = t.b.y_end - t.b.y_begin;
    if (t.own_rows < 1 || n < 1) return -1;
    t.tiles_x = (nw + (WX - 2 * kTbHaloWords) - 1) / (WX - 2 * kTbHaloWords);
    const long tiles_y = (n + t.own_rows - 1) / t.own_rows;
    const dim3 grid((unsigned)(t.tiles_x * tiles_y)), block(WX * RT);
    if (t.b.ntab == 1)
        hipLaunchKernelGGL((fdtd2d_tile<WX, RT, V, true>), grid, block, 0, stream, t);
    else
        hipLaunchKernelGGL((fdtd2d_tile<WX, RT, V, false>), grid, block, 0, stream, t);
    return 0;
}

} // namespace

// Tile height: 32 rows (512 threads) while taller tiles would leave CUs without a workgroup, and
// again once there are several rounds of them (two workgroups per CU then cover each other's
// barriers: 2048^2 311 against 300 Gcell-updates/s); in between 64 rows (1024 threads, fewer
// recomputed rows: 768^2 2.8 against 3.7 us per step).
int sweep2d_tile_rows(int nx, int rows_of_launch, int steps)
{
    static const int forced = std::getenv("FDTD_TILE_ROWS") ? std::atoi(std::getenv("FDTD_TILE_ROWS")) : 0;
    if (forced) return forced;
    const long nw = (nx + 3) / 4, n = rows_of_launch;
    const long tall = ((nw + 11) / 12) * ((n + (64 - 2 * steps) - 1) / (64 - 2 * steps));
    return (tall >= 256 && tall <= 1100) ? 64 : 32; // (1536^2, 1024 tall tiles: 268 against 235; 1792^2, 1444: 282 against 291)
}

int launch_sweep2d_tile(const SweepTile &t, void *stream)
{
    if (t.steps < 1 || t.steps > kMaxStepsPerPass) return -1;
    switch (sweep2d_tile_rows(t.b.g.nx, t.b.y_end - t.b.y_begin, t.steps)) {
    case 32: return launch_tile<16, 32, 1>(t, (hipStream_t)stream);
    case 64: return launch_tile<16, 64, 1>(t, (hipStream_t)stream);
    case 642: return launch_tile<16, 32, 2>(t, (hipStream_t)stream); // 64 rows, two words per thread
    default: return -1;
    }
}

const char *sweep2d_tile_kernel_name() { return "fdtd2d_tile"; }

} // namespace fdtd
