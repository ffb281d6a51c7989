// fdtd2d_sweep_tb: K time steps of the 2-D TMz grid in ONE pass over memory.
//
// A one-step sweep has to move every live array through HBM once per step
// (41 bytes per cell).  Here a wave carries a window of rows through K chained
// stages in registers -- stage s applies updateHField + updateDField +
// updateSources + updateEField (fdtd_calculations.cu:32-213) and the Mur update
// (fdtd_calculations.cu:369-427) of step n+s to a row as soon as stage s-1 has
// produced the rows it depends on -- so a pass reads levels n, n-1 and H once
// (25 bytes per cell) and writes levels n+K, n+K-1 and H once (24 bytes per
// cell) whatever K is.  The arithmetic per cell and step is the same sequence
// of IEEE operations as in the one-step sweep.
//
// Work decomposition.  A wave owns 60 consecutive 16-byte words (240 columns,
// 960 bytes: every wave's stores start on a 64-byte boundary, which HBM writes
// need -- 62 owned words cost 6 %) of `rows` consecutive rows; lanes 0-1 and
// 62-63 hold the words to the left and right of them, whose level-s values are
// only valid in the 8 - s columns next to the owned words (one column is lost
// per step, hence K <= 8; registers and LDS are full at K = 8 too).  A launch
// covers a range of rows (a pass may be split into a launch for the rows next
// to the slab edges and one for the rest, so that the halo exchange of the
// former overlaps the latter).  The wave marches upwards; in iteration j stage
// s works on row j - (s - 1), so the pipeline is primed K rows below the first
// owned row and drained K - 1 rows above the last: rows outside the owned
// window are recomputed, not communicated.  While the pipeline fills, only the
// stages whose rows hold valid data run (see the march below).  Because
// neighbouring waves re-read the rows a wave writes, the
// output levels live in arrays of their own (no in-place update), and H has a
// read and a write generation.
//
// State.  Between iterations a wave keeps, per stage, ez, hx, dz of the level
// below in registers; what is read exactly once -- hy of that level, ez and dz
// two levels below -- waits in LDS (one slot per lane: no barriers).  Waves whose cells
// all lie inside every guard and away from faces, sources and probes take a
// copy of the loop in which the guards are folded away at compile time.
//
// Mur.  Levels 1 .. K-1 get their boundary values in registers: the x faces
// right after the E update of the row, the y = 0 row one iteration later (it
// needs the new row 1), the y = ny-1 row from the carried row ny-2.  Level K is
// stored as the E update leaves it; the ordinary Mur kernel finishes it.
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "device_math.h"

namespace fdtd {

namespace {

// Level s-1 of one row: what stage s consumes and a wave carries in registers from one iteration
// to the next.  Stage s also needs hy of level s-1 and ez, dz of level s-2, each read exactly once:
// those wait in LDS.
struct Level {
    f4 ez, hx, dz;
    unsigned cd; // four material codes
};

struct StageOut {
    f4 ez, hx, hy, dz;
};

// f(integral_constant<int, 1>), f(integral_constant<int, 2>), ... one call per element of the sequence
template <int... N, class F> __device__ __forceinline__ void for_each_count(std::integer_sequence<int, N...>, F &&f)
{
    (f(std::integral_constant<int, N + 1>{}), ...);
}

// Which guards a copy of the marching loop folds away at compile time.
template <bool INX, bool INY> struct LoopMode {
    static constexpr bool inx = INX, iny = INY;
};

// 1: the first 2(K-1) iterations of a wave run only the stages that have valid rows; 0: all K stages.
#ifndef FDTD_TB_SKIP_FILL
#define FDTD_TB_SKIP_FILL 1
#endif
// Waves per workgroup: four, except for the 16-byte-row table variant at K = 8, whose 20 KB of parked
// rows per wave plus the 4 KB table allow three (two workgroups per CU either way).  The 48-byte-row
// variant holds at most kWideTableRowsK8 rows there and keeps one more hy row in registers, which
// leaves room for four waves: 2 x (4 x 19 KB + 4 KB) = 160 KB.
#ifndef FDTD_TB8_TABLE_WAVES
#define FDTD_TB8_TABLE_WAVES 3
#endif
template <int K, bool ONE, bool WIDE> constexpr int tb_waves() { return (!ONE && !WIDE && K >= 8) ? FDTD_TB8_TABLE_WAVES : 4; }
template <int K> constexpr int wide_rows() { return K >= 8 ? kWideTableRowsK8 : kWideTableRows; }

// At least two waves per SIMD (the register allocator otherwise settles for one at some K).
// PLAIN: loads without the nontemporal hint (the default).
// WIDE (table variant): rows of 48 bytes that spell out the three fp64 factors of the E update,
// which the 16-byte rows make every cell rebuild from dt and tau (7 fp64 instructions per cell
// and stage); at most kWideTableRows material codes.
template <int K, bool ONE, bool PLAIN, bool WIDE>
__global__ __launch_bounds__((tb_waves<K, ONE, WIDE>() * 64), 2) void fdtd2d_sweep_tb(const SweepTB a)
{
    constexpr int OWN = kTbOwnWords, LEFT = kTbHaloWords, WPB = tb_waves<K, ONE, WIDE>();
    if (a.nwaves == 0) return; // the launch that loads the kernel when a context is created
    // Grids with several materials: the coefficient table sits in LDS, where a lookup per cell and
    // stage costs a fraction of a cached global load.
    __shared__ MatEntry tab[(ONE || WIDE) ? 1 : FDTD_MAX_CODES];
    __shared__ MatCoef tabw[(!ONE && WIDE) ? wide_rows<K>() : 1];
    if (!ONE) {
        if (WIDE) {
            const f4 *src = (const f4 *)a.table_wide;
            f4 *dst = (f4 *)tabw;
            for (int i = threadIdx.x; i < wide_rows<K>() * 3; i += WPB * 64) dst[i] = src[i];
        } else {
            for (int i = threadIdx.x; i < FDTD_MAX_CODES; i += WPB * 64) tab[i] = a.table[i];
        }
        __syncthreads(); // before any wave leaves
    }
    const Grid2D &g = a.g;
    const int lane = threadIdx.x & 63;
    // wave timeline (diagnostics, a.trace != nullptr): per launch position four words -- start and end
    // on the 100 MHz clock, strip | band << 32, loop copy | hardware ids
    const long trace_pos = (long)blockIdx.x * WPB + (threadIdx.x >> 6);
    const unsigned long long trace_t0 = a.trace ? wall_clock64() : 0ull;
    int strip, band;
    if (a.map == 0) {
        long wv = (long)blockIdx.x * WPB + (threadIdx.x >> 6);
        // positions past the last wave of the launch (the last block is padded to WPB waves): the
        // remaps below would turn them into waves that exist
        if (wv >= a.nwaves) return;
        if (a.nfirst > 0) {
            // the waves listed in first[] (ascending) go first, the others follow in their order
            if (wv < a.nfirst) {
                wv = a.first[wv];
            } else {
                wv -= a.nfirst;
                for (int i = 0; i < a.nfirst; ++i)
                    if (a.first[i] <= wv) ++wv;
            }
        }
        if (a.faces_first) {
            // the two strips on the x faces of every band (guarded loop: slow waves) before the others
            const long nb = a.faces_first; // bands of the launch
            if (wv < 2 * nb) {
                strip = (wv & 1) ? a.strips - 1 : 0;
                band = (int)(wv >> 1);
            } else {
                const long q = wv - 2 * nb;
                strip = 1 + (int)(q % (a.strips - 2));
                band = (int)(q / (a.strips - 2));
            }
            // and the last band (the y face at the upper end: guarded too) right after the first
            if (nb >= 3) band = band == 1 ? (int)nb - 1 : band > 1 ? band - 1 : 0;
        } else {
            strip = (int)(wv % a.strips);
            band = (int)(wv / a.strips);
        }
    } else {
        // Block ids are dealt to the 8 XCDs in turn: XCD i (block id mod 8) walks panel i, a run of
        // strips_per_xcd adjacent strips, band after band, so the lines two adjacent strips share are
        // fetched by one L2.  (Placement is the hardware's habit, not a promise: it only affects speed.)
        const unsigned xcd = blockIdx.x & 7u;
        const long q = (long)(blockIdx.x >> 3) * WPB + (threadIdx.x >> 6);
        band = (int)(q / a.strips_per_xcd);
        strip = (int)xcd * a.strips_per_xcd + (int)(q % a.strips_per_xcd);
        if (strip >= a.strips) return;
    }
    // Bands: a head band on the y = 0 face, big_bands bands of `rows` rows, bands of tail_rows rows
    // (short waves at the end of the launch keep the GPU full while the last long ones finish: 3 % at
    // 16384^2), a last band on the y = ny-1 face.
    // (one strip and one band per wave: said out loud, the row arithmetic, the guards in y and the
    // row part of every address go to the scalar unit)
    strip = __builtin_amdgcn_readfirstlane(strip);
    band = __builtin_amdgcn_readfirstlane(band);
    int y0, y1;
    if (!tb_band_rows(a, band, y0, y1)) return;
    const int w = strip * OWN - LEFT + lane; // word of the row held by this lane
    const int x4 = w * 4;
    const bool act = w >= 0 && x4 < g.nx;
    const bool own = act && lane >= LEFT && lane < LEFT + OWN;
    const long pitch = g.pitch;

    bool m_hx[4], m_hy[4], m_ez[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int x = x4 + i;
        m_hx[i] = x >= 1 && x <= g.nx - 2; // also the dz guard and the x range of the y-face Mur cells
        m_hy[i] = x >= 0 && x <= g.nx - 2;
        m_ez[i] = x >= 1 && x <= g.nx - 1;
    }
    // the x = nx-1 Mur cell: word and component, and whether this wave holds it
    const int xe_w = (g.nx - 1) >> 2, xe_c = (g.nx - 1) & 3;
    const int w_first = strip * OWN - LEFT;
    const bool wave_xe = xe_w >= w_first && xe_w < w_first + 64;

    // sources and probes inside the window of this wave (wave-uniform masks)
    const int jlo = y0 - K, jhi = y1 + K - 2;
    unsigned srcmask = 0, prbmask = 0;
    for (int s = 0; s < a.src.count; ++s) {
        const int js = a.src.y[s] - g.gy0, ws = a.src.x[s] >> 2;
        if (ws >= w_first && ws < w_first + 64 && js >= jlo && js <= jhi) srcmask |= 1u << s;
    }
    if (K > 1 && a.ring) {
        for (int s = 0; s < a.probes.count; ++s) {
            const int js = a.probes.y[s] - g.gy0, ws = a.probes.x[s] >> 2;
            if (ws >= w_first + LEFT && ws < w_first + LEFT + OWN && js >= y0 && js < y1 && a.probes.x[s] >= 0 &&
                a.probes.x[s] < g.nx)
                prbmask |= 1u << s;
        }
    }

    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    const OneCoef onec = one_coef(a.one, g.dt); // (used by the single-material variants only)

    // LDS: per wave and stage s >= 2 one slot {ez, dz of level s-2, hy of level s-1} for the row
    // stage s works on next; every lane reads and writes its own 16 bytes, no other wave touches
    // them: no barriers.
    // (K = 8: 21 slots per wave would leave room for one block per CU only; the hy of its last
    // stage -- with the 48-byte table rows, of its last two stages -- stays in registers instead.)
    // Slots of a wave: {ez, dz} of stage 2, 3, ... K first, then the hy slots of the stages that park it.
    constexpr int HYR = K >= 8 ? ((!ONE && WIDE) ? 2 : 1) : 0;
    __shared__ f4 parked[WPB][K > 1 ? 3 * (K - 1) - HYR : 1][64];
    f4(*park_w)[64] = parked[threadIdx.x >> 6];
    auto park = [&](int stage_index, int which) -> f4 & {
        return park_w[which < 2 ? 2 * stage_index + which : 2 * (K - 1) + stage_index][lane];
    };

    // One stage: level s-1 of row r (with ez, dz of level s-2, ez of row r + 1, hx of level s of
    // row r - 1) -> level s of row r.
    // `inner` (a compile-time flag): the wave and all its rows lie inside the guards of every
    // update, away from the Mur faces, sources and probes -- the guards fold away.
    auto stage = [&](auto mode, const Level &in, const f4 &hyin, const f4 &ezp, const f4 &dzp, const f4 &ezu,
                     const f4 &hxb, int r, float jz) -> StageOut {
        // INX: every lane holds cells 1 .. nx-2 only; INY: every row any stage touches lies in 2 .. ny-3
        // and the wave holds no source and no probe.  The guards of a folded direction vanish.
        constexpr bool INX = decltype(mode)::inx, INY = decltype(mode)::iny;
        const int y = g.gy0 + r;
        const bool gy_hx = y >= 0 && y <= g.gny - 2;
        const bool gy_in = y >= 1 && y <= g.gny - 2;
        const bool gy_ez = y >= 0 && y <= g.gny - 1;
        StageOut o;
        // the four table rows of this lane's cells first: the E update at the end of the stage needs
        // them, and the H and D arithmetic in between covers the LDS latency
        // (wide rows: when the four cells of every lane of the wave hold one code each -- always so
        // on a map upsampled by a multiple of four -- one row per lane serves all four cells; a
        // wave with a mixed word anywhere reads a row per cell just before it is used)
        MatEntry m[4];
        MatCoef mu;
        bool words_uniform = false;
        if (!ONE && WIDE) {
            const unsigned c0 = in.cd & 0xffu;
            words_uniform = __all(in.cd == c0 * 0x01010101u);
            mu = tabw[c0];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) m[i] = ONE ? a.one : tab[(in.cd >> (8 * i)) & 0xffu];
        }
        const float e_right = __shfl_down(in.ez.x, 1);
        const float er[4] = {in.ez.y, in.ez.z, in.ez.w, e_right};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o.hx[i] = ((INY || gy_hx) && (INX || m_hx[i])) ? in.hx[i] - g.kh_y * (ezu[i] - in.ez[i]) : in.hx[i];
            o.hy[i] = ((INY || gy_in) && (INX || m_hy[i])) ? hyin[i] + g.kh_x * (er[i] - in.ez[i]) : hyin[i];
        }
        const float hy_left = __shfl_up(o.hy.w, 1);
        const float hl[4] = {hy_left, o.hy.x, o.hy.y, o.hy.z};

        unsigned srcflag = 0;
        if (!INY && srcmask) {
            for (int s = 0; s < a.src.count; ++s) {
                if (!((srcmask >> s) & 1u) || a.src.y[s] - g.gy0 != r) continue;
                if ((a.src.x[s] >> 2) == w) srcflag |= 1u << (a.src.x[s] & 3);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v = (in.dz[i] + g.cx * (o.hy[i] - hl[i])) - g.cy * (o.hx[i] - hxb[i]);
            float dn = ((INY || gy_in) && (INX || m_hx[i])) ? v : dzp[i];
            if (!INY && ((srcflag >> i) & 1u)) dn = v - jz;
            o.dz[i] = dn;
        }

        // (the new value is computed for every cell and the guard picks: written as `guard ? update :
        // old` the compiler branches around each cell's fp64 chain, and the four chains of a lane, no
        // longer interleaved, take the strips on the x faces half as long again as the others)
        f4 en;
        if (!ONE && WIDE) {
            if (words_uniform) {
#pragma unroll
                for (int i = 0; i < 4; ++i) en[i] = e_update_wide(mu, in.ez[i], ezp[i], o.dz[i], in.dz[i], dzp[i]);
            } else {
                // (a word with several materials: a table row of twelve registers per cell, taken when
                // the cell's turn comes -- all four at once do not fit beside the state of the march)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    en[i] = ezp[i];
                    if ((INY || gy_ez) && (INX || m_ez[i])) {
                        const MatCoef mi = tabw[(in.cd >> (8 * i)) & 0xffu];
                        en[i] = e_update_wide(mi, in.ez[i], ezp[i], o.dz[i], in.dz[i], dzp[i]);
                    }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                en[i] = ONE ? e_update_one(onec, in.ez[i], ezp[i], o.dz[i], in.dz[i], dzp[i])
                            : e_update(m[i], g.dt, in.ez[i], ezp[i], o.dz[i], in.dz[i], dzp[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) o.ez[i] = ((INY || gy_ez) && (INX || m_ez[i])) ? en[i] : ezp[i];
        return o;
    };

    // Mur on the x faces of a freshly computed row (level s), fdtd_calculations.cu:369-395.
    // Its field-independent factors come from the per-row arrays.  A wave holds one x face (grids
    // narrower than a strip: both, see below); the factors of the K rows its stages work on wait in
    // ONE register quadruple spread over the lanes -- lane l holds those of row j - l in iteration j,
    // the entry of row j + 1 arrives with that row's loads and the queue moves up a lane per
    // iteration --, and stage s takes its row's entry out of lane s - 1.  (A load per stage, as this
    // was first written, made the strips on the x faces take twice as long as the others: the load
    // had not arrived when the stage was done, and waiting for it waits for the next row's loads too.)
    const bool face_x0 = w_first <= 0;              // the wave holds word 0
    const bool face_both = face_x0 && wave_xe;      // grid of less than one strip: the x0 factors are loaded per stage
    auto mur_x_row = [&](int j) -> f4 {             // the queue's entry for row j
        // (rows outside 1 .. gny-2 take no Mur update and a wave without a face uses none: any entry
        // the slab holds will do, and an unconditional load needs no wait where a branch would join)
        const int y = min(max(g.gy0 + j, 1), g.gny - 2);
        return ((const f4 *)(wave_xe ? a.cfxe : a.cfx0))[y];
    };
    // (Every lane evaluates the face formula and the face lane takes the result: a select costs less
    // than a divergent branch per stage, and the cell next to x = nx-1 sits in the same word unless
    // nx = 4m + 1, so the two lane exchanges are rarely needed.)
    auto mur_x = [&](f4 &et, const f4 &es, int r, const f4 &cf) {
        const int y = g.gy0 + r;
        const bool yok = y >= 1 && y <= g.gny - 2;
        if (face_x0) {
            f4 c0 = cf;
            if (face_both) c0 = ((const f4 *)a.cfx0)[min(max(y, 1), g.gny - 2)];
            const float v = mur_apply(c0, et.y, es.y, es.x);
            et.x = (w == 0 && yok) ? v : et.x;
        }
        if (wave_xe) {
            const bool on = w == xe_w && yok;
            if (xe_c == 0) {
                const float t_prev = __shfl_up(et.w, 1), s_prev = __shfl_up(es.w, 1);
                const float v = mur_apply(cf, t_prev, s_prev, es.x);
                et.x = on ? v : et.x;
            } else if (xe_c == 1) {
                const float v = mur_apply(cf, et.x, es.x, es.y);
                et.y = on ? v : et.y;
            } else if (xe_c == 2) {
                const float v = mur_apply(cf, et.y, es.y, es.z);
                et.z = on ? v : et.z;
            } else {
                const float v = mur_apply(cf, et.z, es.z, es.w);
                et.w = on ? v : et.w;
            }
        }
    };
    // Mur on the y faces, fdtd_calculations.cu:401-427: `here` is the face row, `in` the row next to it.
    // (factors per column from the arrays: spelled out -- two square roots and a division per cell --
    // the two cases of every stage made the guarded loop eight times the size of the plain one)
    auto mur_y = [&](f4 &et_here, const f4 &et_in, const f4 &es_in, const f4 &es_here, const float *cf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (!m_hx[i]) continue;
            et_here[i] = mur_apply(((const f4 *)cf)[x4 + i], et_in[i], es_in[i], es_here[i]);
        }
    };

    auto record = [&](const Level &v, const f4 &vhy, int r, int slot) {
        for (int s = 0; s < a.probes.count; ++s) {
            if (!((prbmask >> s) & 1u) || a.probes.y[s] - g.gy0 != r || (a.probes.x[s] >> 2) != w) continue;
            const int i = a.probes.x[s] & 3;
            float *dst = a.ring + ((size_t)slot * FDTD_MAX_PROBES + s) * 4;
            float hx = v.hx.x, hy = vhy.x, dz = v.dz.x, ez = v.ez.x;
#pragma unroll
            for (int q = 1; q < 4; ++q)
                if (q == i) hx = v.hx[q], hy = vhy[q], dz = v.dz[q], ez = v.ez[q];
            dst[0] = hx;
            dst[1] = hy;
            dst[2] = dz;
            dst[3] = ez;
        }
    };

    // Everything a wave loads for one row (levels n, n-1 and H), and ez^n of the row above.
    struct RowIn {
        f4 hx, hy, ezp, dz, dzp, ezu;
        f4 cfx; // x-face Mur factors of the row (waves on an x face only)
        unsigned cd;
    };
    auto ld4 = [&](const float *p) -> f4 { return PLAIN ? *(const f4 *)p : ld4_stream(p); };
    // Lanes outside the row (left of word 0, right of the last word) load the nearest word of the row
    // instead of nothing: what they hold is never stored and never reaches a cell inside the grid (the
    // x guards of the updates keep the cells at x = 0 and x = nx-1 from reading beyond them), and loads
    // without a predicate are not waited for one by one.
    const int x4l = act ? x4 : w < 0 ? 0 : ((g.nx - 1) >> 2) << 2;
    auto load = [&](auto mode, RowIn &r, int j) {
        // (the row's address in scalar registers, the lane's 32-bit byte offset beside it)
        const unsigned xb = 4u * (unsigned)(decltype(mode)::inx ? x4 : x4l);
        const long off = (long)(j + 1) * pitch;
        // (the empty asm keeps the compiler from folding the lane offset into twelve per-lane base
        // addresses that it then holds in 24 registers across the loop)
        auto at = [&](const float *row) {
            unsigned long long u = (unsigned long long)row;
            asm volatile("" : "+s"(u));
            return (const float *)((const __attribute__((address_space(1))) char *)u + xb);
        };
        r.hx = ld4(at(a.hx_r + off));
        r.hy = ld4(at(a.hy_r + off));
        r.ezp = ld4(at(a.ez_p + off));
        r.dz = ld4(at(a.dz_n + off));
        r.dzp = ld4(at(a.dz_p + off));
        r.ezu = ld4(at(a.ez_n + off + pitch));
        // (a single material: no code is looked at)
        r.cd = !ONE ? *(const unsigned *)(a.code + (long)(j + 1) * g.cpitch + (xb >> 2)) : 0u;
        if (!decltype(mode)::inx) r.cfx = mur_x_row(j);
    };

    auto march = [&](auto mode) {
    constexpr bool INX = decltype(mode)::inx, INY = decltype(mode)::iny;
    // c[s]: level s-1 of the row stage s works on next (c[1] comes from memory).  While stage s
    // works on row r, c[s + 1] still holds level s of row r - 1: its hx is the hx below that the
    // D update of stage s needs; the last stage keeps that row's hx in hx_top.
    Level c[K + 1];
    f4 hx_top = zero;
    // hy of level s-1 for the last HYR stages (index K - s), this and the next iteration
    f4 hy_reg[HYR > 0 ? HYR : 1], hy_reg_new[HYR > 0 ? HYR : 1];
#pragma unroll
    for (int i = 0; i < (HYR > 0 ? HYR : 1); ++i) hy_reg[i] = hy_reg_new[i] = zero;
#pragma unroll
    for (int s = 0; s <= K; ++s) {
        c[s].ez = c[s].hx = c[s].dz = zero;
        c[s].cd = 0;
    }
    RowIn cur;
    load(mode, cur, jlo);
    f4 ezc = ld4(a.ez_n + (long)(jlo + 1) * pitch + (INX ? x4 : x4l));
    float cfq[4] = {0.f, 0.f, 0.f, 0.f}; // the x-face Mur factors of rows j, j-1, ... in lanes 0, 1, ...

    // One iteration of the march with the first NACT stages at work.
    auto iterate = [&](auto nact_c, int j) {
        constexpr int NACT = decltype(nact_c)::value;
        RowIn nxt;
        c[1].ez = ezc;
        c[1].hx = cur.hx;
        c[1].dz = cur.dz;
        c[1].cd = cur.cd;
        if (!INX) {
            const float in_q[4] = {cur.cfx.x, cur.cfx.y, cur.cfx.z, cur.cfx.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float up = __shfl_up(cfq[i], 1);
                cfq[i] = lane == 0 ? in_q[i] : up;
            }
        }
        f4 ezu = cur.ezu;
        // ez, dz of level s-1 and the new hy of the row stage s has just done: what stage s+1 needs
        // for that row in the next iteration.  They go into its slot once it has taken out this
        // iteration's values.
        f4 pend_e = zero, pend_d = zero, pend_h = zero;
        Level keep[K + 1];
#pragma unroll
        for (int s = 1; s <= K; ++s) {
            if (s > NACT) {
                if (s < K) keep[s + 1] = c[s + 1];
                continue;
            }
            const int r = j - (s - 1);
            f4 ezp, dzp, hyin;
            if (s == 1) {
                ezp = cur.ezp;
                dzp = cur.dzp;
                hyin = cur.hy;
            } else {
                ezp = park(s - 2, 0);
                dzp = park(s - 2, 1);
                park(s - 2, 0) = pend_e;
                park(s - 2, 1) = pend_d;
                if (s > K - HYR) {
                    hyin = hy_reg[s > K - HYR ? K - s : 0];
                    hy_reg_new[s > K - HYR ? K - s : 0] = pend_h;
                } else {
                    hyin = park(s - 2, 2);
                    park(s - 2, 2) = pend_h;
                }
            }
            StageOut o = stage(mode, c[s], hyin, ezp, dzp, ezu, s < K ? c[s < K ? s + 1 : s].hx : hx_top, r,
                               a.jz[s - 1]);
            // the next row's loads go out once stage 1 has consumed this row's registers
            if (s == 1 && j < jhi) load(mode, nxt, j + 1);
            if (s < K) {
                const int y = g.gy0 + r;
                if (!INX) {
                    const f4 cfx = {__int_as_float(__builtin_amdgcn_readlane(__float_as_int(cfq[0]), s - 1)),
                                    __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cfq[1]), s - 1)),
                                    __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cfq[2]), s - 1)),
                                    __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cfq[3]), s - 1))};
                    mur_x(o.ez, c[s].ez, r, cfx);
                }
                if (!INY && (y == g.gny - 1 || y == 1)) {
                    // ez of level s-1 of row r - 1 is parked in the slot of stage s + 1
                    const f4 below = park(s - 1, 0);
                    if (y == 1)
                        mur_y(c[s + 1].ez, o.ez, c[s].ez, below, a.cfy0);
                    else
                        mur_y(o.ez, c[s + 1].ez, below, c[s].ez, a.cfye);
                }
                // c[s + 1] now holds the final level s of row r - 1
                if (!INY && prbmask && own && r - 1 >= y0 && r - 1 < y1)
                    record(c[s + 1], (s + 1 > K - HYR) ? hy_reg[s + 1 > K - HYR ? K - (s + 1) : 0] : park(s - 1, 2), r - 1, a.slot[s]);
                keep[s + 1].ez = o.ez;
                keep[s + 1].hx = o.hx;
                keep[s + 1].dz = o.dz;
                keep[s + 1].cd = c[s].cd;
                pend_e = c[s].ez;
                pend_d = c[s].dz;
                pend_h = o.hy;
            } else if (own && r >= y0 && r < y1) {
                const long q = (long)(r + 1) * pitch;
                const unsigned xb = 4u * (unsigned)x4;
                auto at = [&](float *row) {
                    unsigned long long u = (unsigned long long)row;
                    asm volatile("" : "+s"(u));
                    return (float *)((__attribute__((address_space(1))) char *)u + xb);
                };
                st4_stream(at(a.hx_w + q), o.hx);
                st4_stream(at(a.hy_w + q), o.hy);
                st4_stream(at(a.dz_b + q), o.dz);
                st4_stream(at(a.ez_b + q), o.ez);
                st4_stream(at(a.dz_a + q), c[K].dz);
                st4_stream(at(a.ez_a + q), c[K].ez);
            }
            if (s == K) hx_top = o.hx;
            ezu = o.ez;
            // keep the scheduler from interleaving the stages: that only lengthens live ranges
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s = 2; s <= K; ++s) c[s] = keep[s];
#pragma unroll
        for (int i = 0; i < (HYR > 0 ? HYR : 1); ++i) hy_reg[i] = hy_reg_new[i];
        ezc = cur.ezu;
        cur = nxt;
    };
    // While the pipeline fills, stage s has nothing to do before iteration jlo + 2s - 2: level s is
    // valid from row jlo + s on (a row goes stale per step), stage s reaches that row in iteration
    // jlo + 2s - 1, and the hx of the row below it, which its D update needs, is the one thing the
    // iteration before has to produce.  So the first 2(K-1) iterations run in pairs with 1, 2, ...
    // K-1 stages: K(K-1) stage executions less of a wave's K(rows + 2K - 1), a fifth of the work of
    // a band of 25 rows.  (A test per stage inside one loop costs that loop its register budget;
    // the fully guarded copy, which few waves run, is large enough as it is and fills as before.)
    int j = jlo;
    if constexpr (FDTD_TB_SKIP_FILL && INY && K > 1) {
        for_each_count(std::make_integer_sequence<int, K - 1>{}, [&](auto n) {
#pragma unroll 1
            for (int q = 0; q < 2; ++q, ++j) iterate(n, j);
        });
    }
    // (The hand-over of a stage's state to the next iteration is a register copy -- 144 moves per
    // row at K = 8 -- that a loop unrolled twice would not need; it needs more than the 256 registers
    // two waves per SIMD leave: 11 scratch operations per pair of rows.)
#pragma unroll 1
    for (; j <= jhi; ++j) iterate(std::integral_constant<int, K>{}, j);
    };

    // Three copies of the loop: guards folded in both directions (most waves), in y only (the strips on
    // the x faces, in bands away from the y faces: otherwise the slowest waves of a launch), none folded.
    const bool inner_x = w_first >= 1 && (w_first + 63) * 4 + 3 <= g.nx - 2;
    const bool inner_y = g.gy0 + jlo - (K - 1) >= 2 && g.gy0 + jhi + 1 <= g.gny - 3 && srcmask == 0 && prbmask == 0;
    if (inner_x && inner_y)
        march(LoopMode<true, true>{});
    else if (inner_y)
        march(LoopMode<false, true>{});
    else
        march(LoopMode<false, false>{});
    if (a.trace && lane == 0) {
        unsigned long long *t = a.trace + 4 * trace_pos;
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        t[0] = trace_t0;
        t[1] = wall_clock64();
        t[2] = (unsigned long long)(unsigned)strip | ((unsigned long long)(unsigned)band << 32);
        t[3] = (unsigned long long)((inner_x ? 1u : 0u) | (inner_y ? 2u : 0u)) | ((unsigned long long)(xcc & 0xfu) << 8) |
               ((unsigned long long)hw << 32);
    }
}

template <int K> int launch_k(const SweepTB &a, hipStream_t stream)
{
    const long bands = tb_band_count(a);
    if (bands <= 0) return 0;
    const bool one = a.ntab == 1;
    static const bool narrow = std::getenv("FDTD_TB_NARROW_TABLE") && std::atoi(std::getenv("FDTD_TB_NARROW_TABLE"));
    const bool wide = !one && a.ntab <= wide_rows<K>() && !narrow;
    const long wpb = one ? tb_waves<K, true, false>() : wide ? tb_waves<K, false, true>() : tb_waves<K, false, false>();
    long blocks;
    if (a.map == 0)
        blocks = (bands * a.strips + wpb - 1) / wpb;
    else
        blocks = 8 * ((bands * a.strips_per_xcd + wpb - 1) / wpb);
    const dim3 grid((unsigned)blocks), block((unsigned)(wpb * 64));
    SweepTB b = a;
    b.nwaves = bands * a.strips;
    // Loads without the nontemporal hint: the lines two adjacent waves share stay in L2 for the
    // second of them (16384^2, K = 8: 3.32 ms per pass against 3.42 with the hint).  The hinted
    // form is kept for K = 8 on a single material as a tuning reference (FDTD_TB_PLAIN_LOADS=0).
    static const bool hinted = std::getenv("FDTD_TB_PLAIN_LOADS") && !std::atoi(std::getenv("FDTD_TB_PLAIN_LOADS"));
    if (one) {
        if (K == 8 && hinted)
            hipLaunchKernelGGL((fdtd2d_sweep_tb<K, true, K != 8, false>), grid, block, 0, stream, b);
        else
            hipLaunchKernelGGL((fdtd2d_sweep_tb<K, true, true, false>), grid, block, 0, stream, b);
    } else if (wide) {
        hipLaunchKernelGGL((fdtd2d_sweep_tb<K, false, true, true>), grid, block, 0, stream, b);
    } else {
        hipLaunchKernelGGL((fdtd2d_sweep_tb<K, false, true, false>), grid, block, 0, stream, b);
    }
    return 0;
}

// The first launch of an instantiation costs a quarter of a millisecond more than the later ones
// (16384^2: 2.92 ms against 2.67), whichever kernels ran before it.  A context pays that when it is
// created: one workgroup of every instantiation its passes can take -- full passes, remainder passes,
// either table form --, which leaves at its first instruction.  (Asking for the kernels' attributes
// instead loads the code object but leaves the first launch of each as slow as before.)
// FDTD_TB_PRELOAD=0 turns this off: a profiler's per-kernel averages then hold real passes only.
template <int K> void preload_k(hipStream_t stream)
{
    SweepTB z = {};
    hipLaunchKernelGGL((fdtd2d_sweep_tb<K, true, true, false>), dim3(1), dim3(tb_waves<K, true, false>() * 64), 0, stream, z);
    hipLaunchKernelGGL((fdtd2d_sweep_tb<K, false, true, true>), dim3(1), dim3(tb_waves<K, false, true>() * 64), 0, stream, z);
    hipLaunchKernelGGL((fdtd2d_sweep_tb<K, false, true, false>), dim3(1), dim3(tb_waves<K, false, false>() * 64), 0, stream, z);
}

} // namespace

void preload_sweep2d_tb(void *stream)
{
    static const bool off = std::getenv("FDTD_TB_PRELOAD") && !std::atoi(std::getenv("FDTD_TB_PRELOAD"));
    if (off) return;
    hipStream_t s = (hipStream_t)stream;
    preload_k<2>(s), preload_k<3>(s), preload_k<4>(s), preload_k<5>(s), preload_k<6>(s), preload_k<7>(s), preload_k<8>(s);
}

int launch_sweep2d_tb(const SweepTB &a, int steps, void *stream)
{
    switch (steps) {
    case 2: return launch_k<2>(a, (hipStream_t)stream);
    case 3: return launch_k<3>(a, (hipStream_t)stream);
    case 4: return launch_k<4>(a, (hipStream_t)stream);
    case 5: return launch_k<5>(a, (hipStream_t)stream);
    case 6: return launch_k<6>(a, (hipStream_t)stream);
    case 7: return launch_k<7>(a, (hipStream_t)stream);
    case 8: return launch_k<8>(a, (hipStream_t)stream);
    default: return -1; // a single step is the one-step sweep's job
    }
}

const char *sweep2d_tb_kernel_name() { return "fdtd2d_sweep_tb"; }

} // namespace fdtd
