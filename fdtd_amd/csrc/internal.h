// Internal declarations shared by the host code and the HIP kernels of
// libfdtd_amd.so.  Nothing here is part of the C ABI (include/fdtd_amd.h).
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "fdtd_amd.h"

namespace fdtd {

// Material code that stands for the fsigma/feps_s/feps_inf/ftau_d defaults of
// data/input_params (every cell holds it until materials are loaded).
constexpr int kDefaultCode = 255;

// Debye E-update coefficients of one material (fdtd_calculations.cu:141-156),
// evaluated once per material code instead of once per cell per step.  The
// values depend only on (sigma, eps_s, eps_inf, tau_d, dt, eps0), so a table
// row is bit-identical to what the reference recomputes in every thread.
struct alignas(16) MatCoef {
    float a, c, d, pad0;
    double k1, k2;   // 2(dt + tau), 2dt + 4tau
    double k3, pad1; // 2tau
};
static_assert(sizeof(MatCoef) == 48, "MatCoef is read as three 16-byte LDS words");

// What the kernels need of a MatCoef: 16 bytes per material code.
struct alignas(16) MatEntry {
    float a, c, d, tau;
};

// Rows of the wide coefficient table a workgroup of the multi-step sweep can hold in LDS next to
// its parked rows (48 bytes each); grids whose highest material code is below take it.
constexpr int kWideTableRows = 160;
constexpr int kWideTableRowsK8 = 84; // at 8 steps per pass the LDS has 4 KB left for the table: 84 rows of 48 bytes

struct MaterialRow {
    float sigma, eps_s, eps_inf, tau_d;
};

struct SourceList {
    int count;
    int x[FDTD_MAX_SOURCES], y[FDTD_MAX_SOURCES], z[FDTD_MAX_SOURCES];
    float jz; // sample of this step: jz[n - n % 3] (fdtd.cu:172,310,448), 0 past the table
};

struct ProbeList {
    int count;
    int x[FDTD_MAX_PROBES], y[FDTD_MAX_PROBES], z[FDTD_MAX_PROBES];
};

// ---- 2-D TMz ----------------------------------------------------------
// Storage: (nyl + 2) rows of `pitch` floats per array; storage row r holds
// local row j = r - 1, i.e. global row gy0 + j; rows j = -1 and j = nyl are
// halo rows (zero outside the grid).  Every pointer addresses storage row 0.
struct Grid2D {
    int nx, nyl, gy0, gny, pitch;
    int cpitch; // row pitch of the material code array in bytes (the field arrays may be interleaved row by row)
    float dt;
    float kh_x, kh_y; // dt/(mu0*dx), dt/(mu0*dy)   (fdtd_calculations.cu:49,63)
    float cx, cy;     // dt/dx, dt/dy               (fdtd_calculations.cu:92-93)
    int has_lower;    // a neighbour slab exists below (gy0 > 0)
};

struct Sweep2D {
    Grid2D g;
    const float *ez_n;  // E level n
    float *ez_w;        // holds level n-1, receives n+1
    const float *dz_n;
    float *dz_w;
    const float *hx_r, *hy_r; // H read generation
    float *hx_w, *hy_w;       // H write generation (== read generation for the split kernels)
    const uint8_t *code;
    const MatEntry *table;
    MatEntry one;   // table[0], used when the grid holds a single material (ntab == 1)
    int ntab;
    int chunk_rows;
    int row_begin, row_end; // local rows this launch covers (the fused sweep only)
    int variant;    // tuning switch: bit 0 = nontemporal hint on the last touch of re-read lines
    int band;       // block order: 0 = row-major, else rows per band (multiple of 8)
    int nopad;      // row-major order without padding the x block count to a multiple of 8
    SourceList src;
};

// The multi-step 2-D sweep (sweep2d_tb.hip): K time steps in one pass over memory.
constexpr int kMaxStepsPerPass = 8;     // one column of the two halo words on either side goes stale per step
constexpr int kFastStepsPerPassTable = 7; // grids with several materials: the table takes 4 KB of the LDS, so at 8
                                         // steps per pass a block has three waves instead of four
constexpr int kDefaultStepsPerPass = 8; // 16384^2 vacuum: 4: 358, 5: 430, 6: 498, 7: 562, 8: 612 Gcell-updates/s
constexpr int kTbOwnWords = 60; // 16-byte words of a row a wave owns
constexpr int kTbHaloWords = 2; // words it holds on either side of them
constexpr int kTbFirstWaves = 64; // waves a launch can move to its front
struct SweepTB {
    Grid2D g;
    const float *hx_r, *hy_r;   // H, read generation (step n - 1/2)
    float *hx_w, *hy_w;         // H, write generation (step n + K - 1/2)
    const float *ez_n, *ez_p;   // E levels n, n-1
    const float *dz_n, *dz_p;   // D levels n, n-1
    float *ez_a, *ez_b;         // receive E levels n+K-1, n+K (no aliasing with the inputs: the halo
    float *dz_a, *dz_b;         //   rows of a wave are owned -- and overwritten -- by its neighbours)
    const uint8_t *code;
    const MatEntry *table;
    const MatCoef *table_wide;            // the same rows with the three fp64 factors spelled out
    MatEntry one;
    int ntab;
    SourceList src;                       // src.jz is not used
    float jz[kMaxStepsPerPass];           // source sample of each step of the pass
    const float *rpx0, *rpxe, *rpy0, *rpye; // Mur planes, as in Mur2D
    const float *cfx0, *cfxe;               // per global row the four field-independent factors of the x-face Mur update (MurCoefs)
    const float *cfy0, *cfye;               // the same per column for the y faces
    float dx, dy, mu0, eps0;
    ProbeList probes;
    float *ring;                          // probe ring, nullptr = no probes
    int slot[kMaxStepsPerPass];           // ring slots of the intermediate levels 1 .. K-1
    int rows;                             // owned rows per wave
    // Rows this launch covers: big_bands bands of `rows` rows from y_begin on, then bands of
    // tail_rows rows from tail_begin up to y_end (the rows between the two regions, if any, belong
    // to another launch of the same pass).
    int y_begin, y_end, tail_begin;
    int big_bands, tail_rows;
    // A launch that starts on the y = 0 face begins with one band of head_rows rows, one that ends on
    // the y = ny-1 face ends with one of last_rows rows (0: none): the waves of those two bands run the
    // fully guarded copy of the loop, about twice as slow as the others, and with 2K+1 and K+2 rows the
    // bands next to them lie inside every y guard.  The big bands follow the head band, the tail bands
    // stop last_rows short of y_end.
    int head_rows, last_rows;
    // Waves that hold a source or a probe run the guarded copy of the loop for their whole band and
    // take about twice as long: started in the middle of a launch they finish after everybody else.
    // They go first: position i < nfirst of the launch does the work of wave first[i] (ascending), the
    // positions after them the other waves in their order (map 0 only).
    int nfirst;
    int first[kTbFirstWaves];
    // 0, or the number of bands of the launch: waves are then numbered with the two strips on the x
    // faces of every band first (2 * band + side), the inner strips after them band by band.
    int faces_first;
    int strips;                           // waves across a row (kTbOwnWords owned words each)
    unsigned long long *trace;            // diagnostics: per launch position {start, end, strip | band << 32, flags}, nullptr = off
    long long nwaves;                     // bands x strips (set by the launcher): positions of the padded last block beyond it leave
    int map;                              // 0: waves numbered band by band; 1: each XCD walks its own panel of strips
    int strips_per_xcd;                   // map 1: ceil(strips / 8)
};

#if defined(__HIP__) || defined(__HIPCC__)
#define FDTD_HD __host__ __device__
#else
#define FDTD_HD
#endif

// Bands of a launch of the multi-step sweep, in order: head band, big bands, tail bands, last band.
FDTD_HD inline long tb_band_count(const SweepTB &a)
{
    const int ye = a.y_end - a.last_rows;
    const long tail = a.tail_begin < ye ? (ye - a.tail_begin + a.tail_rows - 1) / a.tail_rows : 0;
    return (a.head_rows > 0 ? 1 : 0) + a.big_bands + tail + (a.last_rows > 0 ? 1 : 0);
}

// Rows [y0, y1) of band `band`; false past the last band.
FDTD_HD inline bool tb_band_rows(const SweepTB &a, long band, int &y0, int &y1)
{
    const int ye = a.y_end - a.last_rows;
    if (a.head_rows > 0) {
        if (band == 0) {
            y0 = a.y_begin;
            y1 = a.y_begin + a.head_rows;
            return true;
        }
        --band;
    }
    if (band < a.big_bands) {
        y0 = a.y_begin + a.head_rows + (int)band * a.rows;
        const int stop = a.tail_begin < ye ? a.tail_begin : ye;
        y1 = y0 + a.rows < stop ? y0 + a.rows : stop;
        return y0 < y1;
    }
    band -= a.big_bands;
    const long tail = a.tail_begin < ye ? (ye - a.tail_begin + a.tail_rows - 1) / a.tail_rows : 0;
    if (band < tail) {
        y0 = a.tail_begin + (int)band * a.tail_rows;
        y1 = y0 + a.tail_rows < ye ? y0 + a.tail_rows : ye;
        return true;
    }
    if (band == tail && a.last_rows > 0) {
        y0 = ye;
        y1 = a.y_end;
        return true;
    }
    return false;
}

// The tiled multi-step 2-D sweep (sweep2d_tile.hip): the arguments of a pass of the multi-step
// sweep (b.rows, b.big_bands, b.tail_*, b.strips, b.map are not used; b.y_begin .. b.y_end are
// the rows the launch covers) and the shape of the launch.
struct SweepTile {
    SweepTB b;
    int steps;      // time steps this pass advances: 1 .. kMaxStepsPerPass
    int halo_rows;  // halo rows the arrays hold on either side of the slab (>= steps)
    int slot_last;  // probe ring slot of the last level (b.slot[] holds those of levels 1 .. steps-1)
    int own_rows;   // rows a tile owns: tile height - 2 * steps (set by the launcher)
    int tiles_x;    // tiles across a row (set by the launcher)
};

struct Mur2D {
    Grid2D g;
    int j_begin, j_end; // local rows this launch finishes (x-face cells of those rows, y faces that lie in them)
    float *ez_t;       // level n+1
    const float *ez_s; // level n
    const float *rpx0, *rpxe, *rpy0, *rpye; // global layouts of fdtd.cu:952-977
    float dt, dx, dy, mu0, eps0;
};

// ---- 3-D ----------------------------------------------------------------
// Storage: (nzl + 2) planes of ny rows of `pitch` floats; storage plane q holds
// local plane k = q - 1 (global gz0 + k); planes k = -1 and k = nzl are halos.
struct Grid3D {
    int nx, ny, nzl, gz0, gnz, pitch;
    int rowlen; // floats of one row (the code array's pitch in bytes); pitch = 3 * rowlen when the three
                // components of a field are interleaved row by row, else rowlen
    float dt;
    float kh_x, kh_y, kh_z;
    float cx, cy, cz;
    int has_lower;
};

struct Sweep3D {
    Grid3D g;
    const float *e_n[3];
    float *e_w[3];
    const float *d_n[3];
    float *d_w[3];
    float *h[3];
    const uint8_t *code;
    const MatEntry *table;
    SourceList src;
};

struct Mur3D {
    Grid3D g;
    int k_begin, k_end; // local planes this launch finishes
    float *e_t[3];
    const float *e_s[3];
    const float *rp[6];
    float dt, dx, dy, dz, mu0, eps0;
};

// The fused 3-D sweep (sweep3d.hip): H read and write generations, sources as
// word indices (plane * plane_words + word within the plane).
struct Fused3D {
    Grid3D g;
    const float *e_n[3];
    float *e_w[3];
    const float *d_n[3];
    float *d_w[3];
    const float *h_r[3];
    float *h_w[3];
    const uint8_t *code;
    const MatEntry *table;
    MatEntry one;
    int ntab;
    int nsrc;
    float jz;
    int k_begin, k_end;                    // local planes this launch covers
    long long plane_words;                 // 16-byte words of one plane: ny * pitch / 4
    int flat;                              // planes per segment space: 1, or all planes of the launch (small grids)
    int seg_words;                         // words of one XCD's segment of a space (an eighth of it, rounded up to waves)
    int seg_blocks;                        // blocks per segment
    int block;                             // threads per block: 256, or 64 / 128 where 256 would leave waves idle
    long long src_word[FDTD_MAX_SOURCES];  // word holding the source cell, -1 if another slab holds it
    int src_comp[FDTD_MAX_SOURCES];        // which of the word's four cells
    // The probe row of the step BEFORE this one, read from the arrays this sweep only reads (the
    // state that step left) by one more workgroup at the end of the grid: a launch less per step.
    float *ring;                           // nullptr: nothing to record
    int probe_slot;
    ProbeList probes;
    // Small whole grids: updateMurBoundary inside the sweep, a launch less per step.  The thread that
    // owns the cell next to a face cell finishes that face cell (it holds both time levels of its own
    // cell and E^n of the face cell); the owner of the face cell leaves that component alone.
    int mur;                               // 0: fdtd3d_mur follows the sweep
    const float *cf[6];                    // per face cell the four factors of MurCoefs (16-byte entries)
};

// The field-independent part of the Mur update of one face, taken once when its rp plane changes:
// cf[cell] = mur_coefs(...) of the face cell and the cell one step inside.  Face cells are numbered
// z * ny + y (x faces), z * nx + x (y faces), y * nx + x (z faces).
struct MurCoefs {
    const float *rp;        // indexed with the plane's global layout (the caller holds cells first .. first + count - 1)
    float *cf;              // likewise, by face cell
    long long first, count; // the face cells to do
    int face;   // FDTD_RP_*
    int nx, ny, nz;
    float dt, du, mu0, eps0;
};

// Deep halo exchange of the multi-step mode: `depth` rows of every live array (H, both E and D
// levels, material codes) next to a slab boundary, packed into / unpacked from one buffer per side:
// buf[side] = 6 arrays x depth rows x nx floats, then depth x nx code bytes.
struct HaloDeep {
    float *arr[6];   // hx, hy, ez newest, ez previous, dz newest, dz previous (local row -1 at storage row 0)
    uint8_t *code;
    float *buf[2];   // side 0 = lower neighbour, side 1 = upper neighbour
    int nx, pitch, cpitch, nyl, depth;
    int sides;       // bit mask of the sides to handle
    int unpack;      // 0: owned rows -> buf; 1: buf -> halo rows
};

// Nearest-neighbour upsampling of base images of device codes into the code array: plane k, row j,
// column x takes base[k][(row0 + j) / up][x / up].
struct ExpandCodes {
    uint8_t *code;         // first owned row (2-D) / plane (3-D)
    const uint8_t *base;   // planes x by x bx bytes
    int nx, bx, by, up;
    int rows, row0;        // rows per plane written, global index of the first (2-D slabs: the slab's first row)
    int planes;
    long long row_pitch, plane_pitch; // bytes
};

struct Probe2D {
    Grid2D g;
    const float *hx, *hy, *dz, *ez;
    ProbeList p;
    float *ring;
    int slot;
    int nmid;                            // earlier slots the multi-step sweep filled for the probes held here
    int mid_slot[kMaxStepsPerPass];
};

struct Probe3D {
    Grid3D g;
    const float *f[9];
    ProbeList p;
    float *ring;
    int slot;
};

// Kernel launchers (kernels.hip).  `stream` is a hipStream_t.
void launch_h2d(const Sweep2D &a, void *stream);
void launch_de2d(const Sweep2D &a, void *stream);
void launch_sweep2d(const Sweep2D &a, void *stream);
// Returns 0, or -1 when no kernel exists for `steps`.
int launch_sweep2d_tb(const SweepTB &a, int steps, void *stream);
const char *sweep2d_tb_kernel_name();
// One empty workgroup of every instantiation of the multi-step sweep (their first launch is slow).
void preload_sweep2d_tb(void *stream);
// Returns 0, or -1 when the pass cannot be tiled (steps outside 1 .. kMaxStepsPerPass, no rows).
int launch_sweep2d_tile(const SweepTile &t, void *stream);
// Tile height the launcher takes for a launch over `rows_of_launch` rows: 32, 64, or 642 (64 rows, two words per thread).
int sweep2d_tile_rows(int nx, int rows_of_launch, int steps);
const char *sweep2d_tile_kernel_name();
void launch_sweep3d(const Fused3D &a, void *stream);
void launch_mur2d(const Mur2D &a, void *stream);
void launch_h3d(const Sweep3D &a, void *stream);
void launch_de3d(const Sweep3D &a, void *stream);
void launch_mur3d(const Mur3D &a, void *stream);
void launch_mur_coefs(const MurCoefs &a, void *stream);
void launch_halo_deep(const HaloDeep &a, void *stream);
void launch_expand_codes(const ExpandCodes &a, void *stream);
void launch_probe2d(const Probe2D &a, void *stream);
void launch_probe3d(const Probe3D &a, void *stream);
const char *sweep2d_kernel_name();
const char *sweep3d_kernel_name();
const char *de3d_kernel_name();
const char *de2d_kernel_name();

// Host-side set-up (host_setup.cpp).
void set_error(const std::string &msg);
int params_load(const char *path, fdtd_params *out);
void params_derive(fdtd_params *p);
int setup_sources(fdtd_params *p);
MatCoef make_coef(const MaterialRow &m, float dt, float eps0, int ecoef);
float rp_value(const fdtd_params &p, const MaterialRow &m, bool x_face);
int read_material_specs(const char *path, MaterialRow table[FDTD_MAX_CODES], bool present[FDTD_MAX_CODES]);
// The pixel values of one slice file read in sequence into a bx x by image (loadMaterials' token
// semantics, fdtd.cu:920-931).
int read_pgm_base(const char *path, int bx, int by, uint8_t *base);
// Codes of one z slice for rows [y0, y0 + rows) of an nx-wide grid.
int read_pgm_slice(const char *path, int nx, int ny, int upsample, int y0, int rows, uint8_t *dst);

} // namespace fdtd
