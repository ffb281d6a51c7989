// The solver context behind the C ABI (include/fdtd_amd.h): device storage,
// the per-step schedule, field/probe/snapshot transfers.
//
// Storage.  Every field is its own padded array in HBM (SoA): rows of `pitch`
// floats (2-D: a multiple of 64, so every row starts on a 256-byte line; fused
// 3-D sweep: a multiple of 16), one halo row (2-D, y axis) or halo plane (3-D,
// z axis) on either side of the slab -- K halo rows in the multi-step 2-D mode.
// E and D keep two time levels: level n+1 is written over level n-1 once the
// cell's E update has consumed it (the reference rotates three buffers,
// fdtd.cu:206-211; cells it never writes are zero in all three, so two levels
// give the same values).  The multi-step mode holds two such pairs of arrays
// and writes the two newest levels of a pass into the pair it did not read.  Materials are one byte per cell plus a table of at
// most 256 coefficient rows instead of four floats per cell.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "comm.h"
#include "internal.h"

using namespace fdtd;

#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                         \
            return -10;                                                                           \
        }                                                                                         \
    } while (0)

namespace {
constexpr size_t kSlack = 256; // floats (or bytes) of zero padding around each array
constexpr long kFoldMurCells = 2500000;
constexpr long kFlat3dCells = 2500000;  // slabs up to this size: no per-plane segments in the fused 3-D sweep // 3-D grids up to this size take the Mur faces into the sweep

// Staging memory of fdtd_halo_get_noalloc(): lives in the code object, so the entry point never
// allocates.  Every GPU the library runs on holds its own copy (4 MiB) at its own address; the
// name below is only a handle for hipGetSymbolAddress -- host code uses the per-device pointers.
constexpr size_t kScratchFloats = (size_t)1 << 20;
constexpr int kMaxDevices = 64;
static __device__ __align__(256) float scratch[kScratchFloats];
float *g_scratch[kMaxDevices] = {};
std::mutex g_scratch_mutex; // contexts may be created from several host threads

// Once per device, with that device current.  A failure is reported by fdtd_halo_get_noalloc(),
// the only user: every other entry point works without the array.
int scratch_init(int device)
{
    if (device < 0 || device >= kMaxDevices) {
        set_error("device ordinal out of range");
        return -3;
    }
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch[device]) return 0;
    float *ptr = nullptr;
    hipError_t e = hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr) {
        set_error(std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
        return -10;
    }
    g_scratch[device] = ptr;
    return 0;
}

float *scratch_of(int device)
{
    if (device < 0 || device >= kMaxDevices) return nullptr;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    return g_scratch[device];
}
}

struct fdtd_ctx {
    fdtd_params p;
    int mode = FDTD_MODE_3D;
    int kernel = FDTD_KERNEL_SPLIT;
    int ecoef = FDTD_ECOEF_REFERENCE;
    int device = 0;
    // stream: the sweeps; side: snapshot copies; xstream (higher priority): the launches for the
    // rows next to a neighbour slab and the halo exchange that follows them
    hipStream_t stream = nullptr, side = nullptr, xstream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool overlap = true; // run the halo exchange beside the sweep of the interior rows

    int nx = 0, ny = 0, nz = 0; // global extents
    int s0 = 0, sn = 0, gn = 0; // slab axis: first global row held, rows held, global rows
    int pitch = 0;     // floats from one row of a field array to the next
    int cpitch = 0;    // bytes from one row of the code array to the next (2-D)
    size_t celems = 0, clead = 0; // the code array's counterparts of elems and lead
    std::vector<std::pair<float *, size_t>> state_bufs; // what a checkpoint holds: (start, floats)
    size_t rows_store = 0; // storage rows per array (halo rows included)
    size_t elems = 0;      // floats per array without the slack
    int chunk_rows = 1;
    int variant = 1;
    int band = 32, nopad = 0;

    std::vector<void *> allocs;
    float *h[3][2] = {};
    float *e[3][4] = {}; // [component][2 * pair + level slot]; the second pair exists in MULTISTEP mode only
    float *d[3][4] = {};
    uint8_t *code = nullptr;
    MatEntry *table = nullptr;
    MatCoef *table_wide = nullptr;
    MatEntry table0 = {};
    // rp planes (layouts of fdtd.cu:952-1001).  Every layout is slowest in the slab axis, so what a
    // slab reads of a plane is ONE run [rp_first, rp_first + rp_n) of the plane's rp_gn entries: the
    // context holds that run only (a 2-D context: nothing of the z faces).  rp[] and mur_cf[] are
    // the starts of the allocations moved back by the run's offset, so that the kernels index them
    // with the global indices of the reference's layouts.
    float *rp_store[6] = {}, *rp[6] = {};
    size_t rp_first[6] = {}, rp_n[6] = {}, rp_gn[6] = {};
    float rp_default[6] = {};
    float *cf_store[6] = {}, *mur_cf[6] = {}; // four Mur factors per face cell, kept in step with rp[]

    int cur = 0;  // slot of the newest E/D level inside the current pair
    int pair = 0; // which pair of E/D arrays holds the two live levels
    int hgen = 0; // index of the newest H generation
    int tb_steps = 1;  // MULTISTEP: time steps per pass (also the halo depth in rows)
    int tb_rows = 32;  // MULTISTEP: owned rows per wave
    bool tile = false; // MULTISTEP storage and schedule, tiled sweep (sweep2d_tile.hip): small grids
    int tb_map = 0;    // MULTISTEP: 1 = every XCD walks its own panel of strips (16384^2: 4.47 ms per pass against 3.42)
    int halo = 1;      // halo rows (2-D) on either side of the slab
    size_t lead = 0;   // floats between the start of an array and local row -1: (halo - 1) * pitch
    int64_t step = 0;

    MaterialRow mats[FDTD_MAX_CODES];
    int ntab = 1;      // 1 + the highest device code on the owned rows; 1 when every owned cell holds ONE material
                       // (the kernels then take its coefficients from their arguments, table0, and read no code)
    int one_code = 0;  // the device code behind table0: 0 (the defaults), or the single code of a whole grid
    int ntab_halo = 1; // the same for the material codes received into the deep halo rows
    int halo_top[2] = {0, 0};
    bool rp_from_codes = false; // x-face rp entries of the deep halo rows follow the codes that arrive there
    bool spp_fixed = false; // steps per pass do not depend on the materials (slabs, or set by the caller)
    int ntab_eff() const { return std::max(ntab, ntab_halo); }
    std::vector<uint8_t> codes_host; // slab cells, x fastest; empty = all default (or kept as codes_base)
    // materials loaded with upsample > 1: the slice images as read (one per owned plane, all rows),
    // expanded on the device; codes_up = 0 when not in use
    std::vector<uint8_t> codes_base;
    int codes_up = 0, base_x = 0, base_y = 0;
    std::vector<float> rp_host[6];
    SourceList src; // global coordinates

    // probes
    ProbeList probes = {};
    float *probe_ring = nullptr;
    int probe_cap = 0;
    int64_t probe_first = 0; // step index of the oldest row the ring may hold

    // per-launch sweep timing
    std::vector<hipEvent_t> prof_ev;
    std::vector<int> prof_steps; // time steps each timed launch advanced
    unsigned long long *wave_trace = nullptr; // fdtd_profile_waves_begin: 4 words per wave of the next multi-step launch
    long wave_trace_cap = 0, wave_trace_n = 0;
    bool wave_trace_armed = false;
    int prof_cap = 0, prof_n = 0;

    // snapshots: two staging sets (compact device copy + pinned host buffer), used in turn, so a new
    // snapshot only waits when two are still on their way
    struct Snap {
        float *dev = nullptr, *host = nullptr;
        size_t count = 0;
        float *dst = nullptr;
        bool pending = false;
        hipEvent_t copied = nullptr; // the compact device copy has been taken
        hipEvent_t done = nullptr;   // the data is in `host`
    } snap[2];
    int snap_next = 0, snap_last = -1;
    hipEvent_t snap_ev = nullptr;
    float *io_stage = nullptr; // contiguous device staging of fdtd_read_field / fdtd_write_field / fdtd_read_window (short rows)
    size_t io_stage_floats = 0;
    int64_t launches = 0;        // sweeps launched so far
    int64_t snap_guard_at = -1;  // the launch with this number must wait for snap[snap_guard].copied
    int snap_guard = 0;

    Comm *comm = nullptr;
    int rank = 0, world = 1;

    // MULTISTEP slabs: deep halos (tb_steps rows of every live array per side)
    float *halo_send[2] = {}, *halo_recv[2] = {};
    bool auto_kernel = false;
    bool interleave3 = false; // 3-D: the three components of a field share one allocation, row by row
    bool fold_mur = false;    // 3-D fused sweep on a small whole grid: the Mur faces inside the sweep
    bool flat3d = false;      // 3-D fused sweep on a small grid: the planes of a launch as one run of words
    bool probe_pending = false; // inside fdtd_step(): the newest step's probe row is still to be written
    bool halo_stale = true; // set by everything that changes the state behind the stepper's back
    bool deep() const { return kernel == FDTD_KERNEL_MULTISTEP; }
    size_t deep_count() const { return (size_t)6 * tb_steps * nx + ((size_t)tb_steps * nx + 3) / 4; }

    bool is2d() const { return mode == FDTD_MODE_2D_TMZ; }
    bool one_pass() const { return kernel == FDTD_KERNEL_FUSED || kernel == FDTD_KERNEL_MULTISTEP; }
    // level 0 = newest, 1 = previous
    float *E(int k, int level) const { return e[k][2 * pair + (level ? cur ^ 1 : cur)]; }
    float *D(int k, int level) const { return d[k][2 * pair + (level ? cur ^ 1 : cur)]; }
    size_t slab_cells() const { return is2d() ? (size_t)nx * sn : (size_t)nx * ny * sn; }
    int field_comp(int field) const { return field % 3; }
};

namespace {

template <class T> int dev_alloc(fdtd_ctx *c, T **out, size_t count_with_slack)
{
    void *ptr = nullptr;
    HIPCHK(hipMalloc(&ptr, count_with_slack * sizeof(T)));
    HIPCHK(hipMemsetAsync(ptr, 0, count_with_slack * sizeof(T), c->stream));
    c->allocs.push_back(ptr);
    *out = (T *)ptr;
    return 0;
}

int alloc_field(fdtd_ctx *c, float **out)
{
    float *base = nullptr;
    int rc = dev_alloc(c, &base, c->elems + 2 * kSlack);
    if (rc) return rc;
    *out = base + kSlack + c->lead;
    c->state_bufs.push_back({base + kSlack, c->elems});
    return 0;
}

Grid2D grid2d(const fdtd_ctx *c)
{
    Grid2D g;
    const fdtd_params &p = c->p;
    g.nx = c->nx;
    g.nyl = c->sn;
    g.gy0 = c->s0;
    g.gny = c->gn;
    g.pitch = c->pitch;
    g.cpitch = c->cpitch;
    g.dt = p.dt;
    g.kh_x = p.dt / (p.mu0 * p.dx);
    g.kh_y = p.dt / (p.mu0 * p.dy);
    g.cx = p.dt / p.dx;
    g.cy = p.dt / p.dy;
    g.has_lower = c->s0 > 0;
    return g;
}

Grid3D grid3d(const fdtd_ctx *c)
{
    Grid3D g;
    const fdtd_params &p = c->p;
    g.nx = c->nx;
    g.ny = c->ny;
    g.nzl = c->sn;
    g.gz0 = c->s0;
    g.gnz = c->gn;
    g.pitch = c->pitch;
    g.rowlen = c->cpitch;
    g.dt = p.dt;
    g.kh_x = p.dt / (p.mu0 * p.dx);
    g.kh_y = p.dt / (p.mu0 * p.dy);
    g.kh_z = p.dt / (p.mu0 * p.dz);
    g.cx = p.dt / p.dx;
    g.cy = p.dt / p.dy;
    g.cz = p.dt / p.dz;
    g.has_lower = c->s0 > 0;
    return g;
}

// Sources in global coordinates; the sample index of step n is n - n % 3.
float jz_of_step(const fdtd_ctx *c, int64_t step)
{
    const int64_t i = step - step % 3;
    return (i >= 0 && i < FDTD_MAX_JZ) ? c->p.jz[i] : 0.0f;
}

SourceList sources_for_step(const fdtd_ctx *c)
{
    SourceList s = c->src;
    s.jz = jz_of_step(c, c->step);
    return s;
}

// On the device a cell holds 0 for the default material and code + 1 for a
// material code of the spec table, so that a grid without loaded materials
// needs one table row and a loaded one only as many rows as it has codes.
inline uint8_t device_code(uint8_t code) { return code == kDefaultCode ? 0 : (uint8_t)(code + 1); }

int upload_table(fdtd_ctx *c)
{
    std::vector<MatEntry> t(FDTD_MAX_CODES);
    std::vector<MatCoef> w(FDTD_MAX_CODES);
    for (int i = 0; i < FDTD_MAX_CODES; ++i) {
        const MatCoef k = make_coef(c->mats[i], c->p.dt, c->p.eps0, c->ecoef);
        t[device_code((uint8_t)i)] = {k.a, k.c, k.d, c->mats[i].tau_d};
        w[device_code((uint8_t)i)] = k;
    }
    c->table0 = t[c->one_code];
    HIPCHK(hipMemcpyAsync(c->table, t.data(), t.size() * sizeof(MatEntry), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->table_wide, w.data(), w.size() * sizeof(MatCoef), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int upload_rp(fdtd_ctx *c, int face)
{
    if (c->rp_n[face] == 0) return 0;
    HIPCHK(hipMemcpyAsync(c->rp_store[face], c->rp_host[face].data(), c->rp_n[face] * sizeof(float),
                          hipMemcpyHostToDevice, c->stream));
    // The 3-D sweep that finishes the faces itself, and the multi-step 2-D sweep (for the intermediate
    // levels of a pass), read the field-independent factors of the Mur update, taken from the plane here,
    // instead of the plane.
    if (c->fold_mur || (c->kernel == FDTD_KERNEL_MULTISTEP && !c->tile && face % 3 != 2)) {
        if (!c->cf_store[face]) {
            int rc = dev_alloc(c, &c->cf_store[face], c->rp_n[face] * 2);
            if (rc) return rc;
            // (an entry of four factors per pair of rp entries: the run starts at cell rp_first / 2)
            c->mur_cf[face] = c->cf_store[face] - c->rp_first[face] * 2;
        }
        MurCoefs m;
        m.rp = c->rp[face];
        m.cf = c->mur_cf[face];
        m.face = face;
        m.first = (long long)(c->rp_first[face] / 2);
        m.count = (long long)(c->rp_n[face] / 2);
        m.nx = c->nx;
        m.ny = c->ny;
        m.nz = c->nz;
        m.dt = c->p.dt;
        m.du = face % 3 == 0 ? c->p.dx : face % 3 == 1 ? c->p.dy : c->p.dz;
        m.mu0 = c->p.mu0;
        m.eps0 = c->p.eps0;
        launch_mur_coefs(m, c->stream);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

// Device address, row pitch and row count of the owned part of a field.
struct FieldView {
    float *ptr;
    size_t rows;
};

int field_view(fdtd_ctx *c, int field, int level, FieldView *v)
{
    if (field < 0 || field >= FDTD_NFIELDS || level < 0 || level > 1) {
        set_error("bad field id or level");
        return -3;
    }
    const int k = field % 3;
    float *base;
    if (field <= FDTD_HZ)
        base = c->h[k][c->hgen];
    else if (field <= FDTD_DZ)
        base = c->D(k, level);
    else
        base = c->E(k, level);
    if (!base) {
        set_error("field is not part of the 2-D TMz mode (hx, hy, dz, ez only)");
        return -3;
    }
    if (c->is2d()) {
        v->ptr = base + (size_t)c->pitch; // skip the lower halo row
        v->rows = (size_t)c->sn;
    } else {
        v->ptr = base + (size_t)c->ny * c->pitch; // skip the lower halo plane
        v->rows = (size_t)c->sn * c->ny;
    }
    return 0;
}

// After the code array changed: `top` = the highest device code on the owned cells, `single` = the
// code every owned cell holds, or -1.  A WHOLE grid of one material, whichever it is, runs the
// single-material kernels with that material's row as their argument; a slab only when the material
// is the default one (its halo rows hold the neighbours' codes, of which only the highest is known).
int set_code_range(fdtd_ctx *c, int top, int single)
{
    const bool whole = c->sn == c->gn;
    c->one_code = (single > 0 && whole) ? single : 0;
    c->ntab = (single == 0 || c->one_code > 0) ? 1 : top + 1;
    return upload_table(c);
}

int upload_codes(fdtd_ctx *c)
{
    // owned rows only; halo rows keep the default code (their E is never updated here)
    const size_t rows = c->is2d() ? (size_t)c->sn : (size_t)c->sn * c->ny;
    uint8_t *dst = c->code + (c->is2d() ? (size_t)c->cpitch : (size_t)c->ny * c->cpitch);
    std::vector<uint8_t> dev(c->codes_host.size());
    int top = 0;
    for (size_t i = 0; i < dev.size(); ++i) {
        dev[i] = device_code(c->codes_host[i]);
        top = std::max(top, (int)dev[i]);
    }
    HIPCHK(hipMemcpy2DAsync(dst, c->cpitch, dev.data(), c->nx, c->nx, rows, hipMemcpyHostToDevice,
                            c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    bool same = !dev.empty();
    for (size_t i = 1; i < dev.size() && same; ++i) same = dev[i] == dev[0];
    return set_code_range(c, top, same ? (int)dev[0] : -1);
}

int use_device(const fdtd_ctx *c)
{
    HIPCHK(hipSetDevice(c->device));
    return 0;
}

// The rows a neighbour needs / provides.  2-D: one ez row.  3-D: one plane of
// ex, ey, ez.  Addresses are in the newest E level.
float *halo_rows(fdtd_ctx *c, int comp, int side, bool halo)
{
    float *base = c->E(comp, 0);
    const size_t unit = c->is2d() ? (size_t)c->pitch : (size_t)c->ny * c->pitch;
    size_t row;
    if (side == 0)
        row = halo ? 0 : 1;
    else
        row = halo ? (size_t)c->sn + 1 : (size_t)c->sn;
    return base + row * unit;
}

void record_probes(fdtd_ctx *c, int mid, bool launch = true);

int ensure_deep_buffers(fdtd_ctx *c)
{
    if (c->halo_send[0]) return 0;
    for (int s = 0; s < 2; ++s) {
        int rc = dev_alloc(c, &c->halo_send[s], c->deep_count());
        if (!rc) rc = dev_alloc(c, &c->halo_recv[s], c->deep_count());
        if (rc) return rc;
    }
    return 0;
}

// Pack the boundary rows of the sides in `sides` into b0 (lower side) / b1 (upper side), or unpack
// those buffers into the halo rows.
int deep_halo_kernel(fdtd_ctx *c, int sides, bool unpack, float *b0, float *b1, hipStream_t st)
{
    HaloDeep a;
    a.arr[0] = c->h[0][c->hgen];
    a.arr[1] = c->h[1][c->hgen];
    a.arr[2] = c->E(2, 0);
    a.arr[3] = c->E(2, 1);
    a.arr[4] = c->D(2, 0);
    a.arr[5] = c->D(2, 1);
    a.code = c->code;
    a.buf[0] = b0;
    a.buf[1] = b1;
    a.nx = c->nx;
    a.pitch = c->pitch;
    a.cpitch = c->cpitch;
    a.nyl = c->sn;
    a.depth = c->tb_steps;
    a.sides = sides;
    a.unpack = unpack ? 1 : 0;
    launch_halo_deep(a, st);
    HIPCHK(hipGetLastError());
    return 0;
}

// Highest material code among the code bytes at the end of a packed deep-halo block.
int top_code(const uint8_t *codes, size_t n)
{
    int top = 0;
    for (size_t i = 0; i < n; ++i) top = std::max(top, (int)codes[i]);
    return top;
}

// The code bytes (depth rows of nx, as stored on the device) that arrived for the halo rows on
// `side`.  They decide the kernel variant; and where the rp planes follow the loaded materials, the
// x-face entries of those rows -- which the slab recomputes, Mur included -- are derived from them.
int note_halo_codes(fdtd_ctx *c, int side, const uint8_t *codes)
{
    const int depth = c->tb_steps, nx = c->nx;
    c->halo_top[side] = top_code(codes, (size_t)depth * nx);
    c->ntab_halo = 1 + std::max(c->halo_top[0], c->halo_top[1]);
    if (!c->rp_from_codes) return 0;
    auto row_of = [&](uint8_t dev) -> const MaterialRow & { return c->mats[dev == 0 ? kDefaultCode : dev - 1]; };
    for (int r = 0; r < depth; ++r) {
        const int y = side == 0 ? c->s0 - depth + r : c->s0 + c->sn + r;
        if (y < 0 || y >= c->gn) continue;
        for (int k = 0; k < 2; ++k) {
            const size_t i = (size_t)y * 2 + k - c->rp_first[FDTD_RP_X0]; // (both x faces hold the same run)
            c->rp_host[FDTD_RP_X0][i] = rp_value(c->p, row_of(codes[(size_t)r * nx + k]), true);
            c->rp_host[FDTD_RP_XEND][i] = rp_value(c->p, row_of(codes[(size_t)r * nx + nx - 2 + k]), true);
        }
    }
    int rc = upload_rp(c, FDTD_RP_X0);
    return rc ? rc : upload_rp(c, FDTD_RP_XEND);
}

// Halo exchange with ranks rank-1 and rank+1 over the communicator, queued on `st`.
// `scan`: the materials may have changed since the last exchange, so look at the codes that
// arrive (this waits for the exchange): a slab whose own cells all hold one material must still
// take the table kernel when its halo rows, which it recomputes, hold another.
int exchange_halos(fdtd_ctx *c, hipStream_t st, bool scan)
{
    if (!c->comm || c->world == 1) return 0;
    const int lower = c->rank - 1, upper = c->rank + 1;
    if (c->deep()) {
        const int sides = (lower >= 0 ? 1 : 0) | (upper < c->world ? 2 : 0);
        int rc = ensure_deep_buffers(c);
        if (!rc) rc = deep_halo_kernel(c, sides, false, c->halo_send[0], c->halo_send[1], st);
        if (rc) return rc;
        rc = comm_group_start(c->comm);
        const int peer[2] = {lower, upper};
        for (int s = 0; s < 2 && rc == 0; ++s) {
            if (!((sides >> s) & 1)) continue;
            rc = comm_send(c->comm, c->halo_send[s], c->deep_count(), peer[s], st);
            if (rc == 0) rc = comm_recv(c->comm, c->halo_recv[s], c->deep_count(), peer[s], st);
        }
        const int rc2 = comm_group_end(c->comm);
        if (rc || rc2) return rc ? rc : rc2;
        c->halo_stale = false;
        if ((rc = deep_halo_kernel(c, sides, true, c->halo_recv[0], c->halo_recv[1], st))) return rc;
        if (scan) {
            const size_t n = (size_t)c->tb_steps * c->nx;
            std::vector<uint8_t> tmp(n);
            c->halo_top[0] = c->halo_top[1] = 0;
            c->ntab_halo = 1;
            for (int s = 0; s < 2; ++s) {
                if (!((sides >> s) & 1)) continue;
                HIPCHK(hipMemcpyAsync(tmp.data(), (const uint8_t *)(c->halo_recv[s] + (size_t)6 * n), n,
                                      hipMemcpyDeviceToHost, st));
                HIPCHK(hipStreamSynchronize(st));
                if ((rc = note_halo_codes(c, s, tmp.data()))) return rc;
            }
        }
        return 0;
    }
    // (interleaved components: the plane of ex, ey and ez is one run of memory, one message)
    const size_t count = c->is2d() ? (size_t)c->nx : (size_t)c->ny * c->pitch;
    const int c0 = c->is2d() ? 2 : 0, c1 = c->interleave3 ? 1 : 3;
    int rc = comm_group_start(c->comm);
    for (int comp = c0; comp < c1 && rc == 0; ++comp) {
        if (lower >= 0) {
            rc = comm_send(c->comm, halo_rows(c, comp, 0, false), count, lower, st);
            if (rc == 0) rc = comm_recv(c->comm, halo_rows(c, comp, 0, true), count, lower, st);
        }
        if (rc == 0 && upper < c->world) {
            rc = comm_send(c->comm, halo_rows(c, comp, 1, false), count, upper, st);
            if (rc == 0) rc = comm_recv(c->comm, halo_rows(c, comp, 1, true), count, upper, st);
        }
    }
    int rc2 = comm_group_end(c->comm);
    return rc ? rc : rc2;
}

// Which sides have a neighbour rank, and whether a step is cut into a launch for the rows
// (planes) next to them -- followed by the halo exchange, both on xstream -- and a launch for
// the interior rows on the main stream.
struct Split {
    bool lower = false, upper = false, on = false;
    int edge = 0; // rows (planes) per edge region
};

Split split_of(const fdtd_ctx *c, int edge, int min_interior)
{
    Split sp;
    if (!c->comm || c->world == 1) return sp;
    sp.lower = c->rank > 0;
    sp.upper = c->rank < c->world - 1;
    sp.edge = edge;
    sp.on = c->overlap && c->one_pass() && (sp.lower || sp.upper) && c->sn >= 2 * edge + min_interior;
    return sp;
}

// A snapshot reads the newest arrays on the side stream while the next sweep runs; the sweep
// after that one overwrites them and has to wait for the copy.
int snapshot_guard(fdtd_ctx *c)
{
    if (c->snap_guard_at >= 0 && c->launches >= c->snap_guard_at) {
        HIPCHK(hipStreamWaitEvent(c->stream, c->snap[c->snap_guard].copied, 0));
        c->snap_guard_at = -1;
    }
    c->launches++;
    return 0;
}

int fork_xstream(fdtd_ctx *c)
{
    HIPCHK(hipEventRecord(c->ev_fork, c->stream));
    HIPCHK(hipStreamWaitEvent(c->xstream, c->ev_fork, 0));
    return 0;
}

int join_xstream(fdtd_ctx *c)
{
    HIPCHK(hipEventRecord(c->ev_join, c->xstream));
    HIPCHK(hipStreamWaitEvent(c->stream, c->ev_join, 0));
    return 0;
}

Mur2D mur2d_args(const fdtd_ctx *c, const Grid2D &g, float *target, const float *source)
{
    Mur2D m;
    m.g = g;
    m.j_begin = 0;
    m.j_end = c->sn;
    m.ez_t = target;
    m.ez_s = source;
    m.rpx0 = c->rp[FDTD_RP_X0];
    m.rpxe = c->rp[FDTD_RP_XEND];
    m.rpy0 = c->rp[FDTD_RP_Y0];
    m.rpye = c->rp[FDTD_RP_YEND];
    m.dt = c->p.dt;
    m.dx = c->p.dx;
    m.dy = c->p.dy;
    m.mu0 = c->p.mu0;
    m.eps0 = c->p.eps0;
    return m;
}

// `last`: no further step follows in this fdtd_step() call.  The fused 3-D sweep records the probe
// row of the step before it, so only the last step of a call needs the probe kernel.
int step_once(fdtd_ctx *c, bool last)
{
    int rc = snapshot_guard(c);
    if (rc) return rc;
    const bool prof = c->prof_n < c->prof_cap;
    // one-step kernels: 8 rows (2-D) or one plane (3-D) next to each neighbour go first
    const Split sp = split_of(c, c->is2d() ? 8 : 1, c->is2d() ? 16 : 2);
    const int lo = sp.on && sp.lower ? sp.edge : 0, hi = sp.on && sp.upper ? c->sn - sp.edge : c->sn;
    if (sp.on && (rc = fork_xstream(c))) return rc;
    if (c->is2d()) {
        Sweep2D a;
        a.g = grid2d(c);
        a.ez_n = c->E(2, 0);
        a.ez_w = c->E(2, 1);
        a.dz_n = c->D(2, 0);
        a.dz_w = c->D(2, 1);
        a.code = c->code;
        a.table = c->table;
        a.ntab = c->ntab;
        a.one = c->table0;
        a.chunk_rows = c->chunk_rows;
        a.row_begin = lo;
        a.row_end = hi;
        a.variant = c->variant;
        a.band = c->band;
        a.nopad = c->nopad;
        a.src = sources_for_step(c);
        Mur2D m = mur2d_args(c, a.g, c->E(2, 1), c->E(2, 0));
        if (c->one_pass()) {
            a.hx_r = c->h[0][c->hgen];
            a.hy_r = c->h[1][c->hgen];
            a.hx_w = c->h[0][c->hgen ^ 1];
            a.hy_w = c->h[1][c->hgen ^ 1];
            if (sp.on) {
                Sweep2D e = a;
                e.band = 0;
                e.nopad = 1;
                const int ranges[2][2] = {{0, lo}, {hi, c->sn}};
                for (const auto &r : ranges) {
                    if (r[1] <= r[0]) continue;
                    e.row_begin = m.j_begin = r[0];
                    e.row_end = m.j_end = r[1];
                    launch_sweep2d(e, c->xstream);
                    launch_mur2d(m, c->xstream);
                }
            }
            if (prof) HIPCHK(hipEventRecord(c->prof_ev[2 * c->prof_n], c->stream));
            launch_sweep2d(a, c->stream);
            c->hgen ^= 1;
        } else {
            a.hx_r = a.hx_w = c->h[0][c->hgen];
            a.hy_r = a.hy_w = c->h[1][c->hgen];
            if (prof) HIPCHK(hipEventRecord(c->prof_ev[2 * c->prof_n], c->stream));
            launch_h2d(a, c->stream);
            launch_de2d(a, c->stream);
        }
        if (prof) HIPCHK(hipEventRecord(c->prof_ev[2 * c->prof_n + 1], c->stream));
        m.j_begin = lo;
        m.j_end = hi;
        launch_mur2d(m, c->stream);
    } else {
        Sweep3D a;
        a.g = grid3d(c);
        for (int k = 0; k < 3; ++k) {
            a.e_n[k] = c->E(k, 0);
            a.e_w[k] = c->E(k, 1);
            a.d_n[k] = c->D(k, 0);
            a.d_w[k] = c->D(k, 1);
            a.h[k] = c->h[k][c->hgen];
        }
        a.code = c->code;
        a.table = c->table;
        a.src = sources_for_step(c);
        Mur3D m;
        m.g = a.g;
        for (int k = 0; k < 3; ++k) {
            m.e_t[k] = c->E(k, 1);
            m.e_s[k] = c->E(k, 0);
        }
        for (int f = 0; f < 6; ++f) m.rp[f] = c->rp[f];
        m.dt = c->p.dt;
        m.dx = c->p.dx;
        m.dy = c->p.dy;
        m.dz = c->p.dz;
        m.mu0 = c->p.mu0;
        m.eps0 = c->p.eps0;
        if (c->kernel == FDTD_KERNEL_FUSED) {
            Fused3D f;
            f.g = a.g;
            for (int k = 0; k < 3; ++k) {
                f.e_n[k] = a.e_n[k];
                f.e_w[k] = a.e_w[k];
                f.d_n[k] = a.d_n[k];
                f.d_w[k] = a.d_w[k];
                f.h_r[k] = c->h[k][c->hgen];
                f.h_w[k] = c->h[k][c->hgen ^ 1];
            }
            f.code = c->code;
            f.table = c->table;
            f.one = c->table0;
            f.ntab = c->ntab;
            f.jz = a.src.jz;
            f.nsrc = a.src.count;
            const long long p4 = c->cpitch / 4;
            f.plane_words = (long long)c->ny * p4;
            // the shape of a launch over local planes [k0, k1)
            auto shape = [&](int k0, int k1) {
                f.k_begin = k0;
                f.k_end = k1;
                f.flat = c->flat3d ? std::max(1, k1 - k0) : 1;
                // wave-aligned segments: the last lane of a segment is lane 63 of its wave
                f.seg_words = (int)(((f.flat * f.plane_words + 7) / 8 + 63) / 64 * 64);
                // 256 threads per block, fewer where the last block of a segment would be mostly idle
                // (100^3: segments of 384 words)
                f.block = f.seg_words % 256 == 0 || f.seg_words >= 4096 ? 256 : f.seg_words % 192 == 0 ? 192 : f.seg_words % 128 == 0 ? 128 : 64;
                f.seg_blocks = (f.seg_words + f.block - 1) / f.block;
            };
            f.ring = nullptr;
            f.probe_slot = 0;
            f.probes = c->probes;
            f.mur = c->fold_mur ? 1 : 0;
            for (int face = 0; face < 6; ++face) f.cf[face] = c->mur_cf[face];
            for (int s = 0; s < a.src.count; ++s) {
                const int kz = a.src.z[s] - c->s0;
                const bool here = kz >= 0 && kz < c->sn;
                f.src_word[s] = here ? ((long long)kz * c->ny + a.src.y[s]) * p4 + a.src.x[s] / 4 : -1;
                f.src_comp[s] = a.src.x[s] % 4;
            }
            if (sp.on) {
                const int ranges[2][2] = {{0, lo}, {hi, c->sn}};
                for (const auto &r : ranges) {
                    if (r[1] <= r[0]) continue;
                    m.k_begin = r[0];
                    m.k_end = r[1];
                    shape(r[0], r[1]);
                    launch_sweep3d(f, c->xstream);
                    launch_mur3d(m, c->xstream);
                }
            }
            shape(lo, hi);
            if (c->probe_pending) {
                f.ring = c->probe_ring;
                f.probe_slot = (int)((c->step - 1) % c->probe_cap);
                c->probe_pending = false;
            }
            if (prof) HIPCHK(hipEventRecord(c->prof_ev[2 * c->prof_n], c->stream));
            launch_sweep3d(f, c->stream);
            c->hgen ^= 1;
        } else {
            if (prof) HIPCHK(hipEventRecord(c->prof_ev[2 * c->prof_n], c->stream));
            launch_h3d(a, c->stream);
            launch_de3d(a, c->stream);
        }
        if (prof) HIPCHK(hipEventRecord(c->prof_ev[2 * c->prof_n + 1], c->stream));
        m.k_begin = lo;
        m.k_end = hi;
        if (!(c->kernel == FDTD_KERNEL_FUSED && c->fold_mur)) launch_mur3d(m, c->stream);
    }
    HIPCHK(hipGetLastError());
    if (prof) c->prof_steps[c->prof_n++] = 1;
    c->cur ^= 1;
    c->step++;

    if (sp.on) {
        if ((rc = exchange_halos(c, c->xstream, false))) return rc;
        if ((rc = join_xstream(c))) return rc;
    } else if ((rc = exchange_halos(c, c->stream, false))) {
        return rc;
    }
    if (!last && !c->is2d() && c->kernel == FDTD_KERNEL_FUSED && c->probes.count > 0 && c->probe_ring) {
        c->probe_pending = true; // the next step's sweep writes the row
        record_probes(c, 0, false);
    } else {
        record_probes(c, 0);
    }
    return 0;
}

// Appends the newest step to the probe ring.  `mid` > 0: the `mid` steps before it were written
// by the multi-step sweep for the probes this slab holds; the others get their zeros here.
void record_probes(fdtd_ctx *c, int mid, bool launch)
{
    if (c->probes.count > 0 && c->probe_ring) {
        const int slot = (int)((c->step - 1) % c->probe_cap);
        if (!launch) {
            // the sweep has written the rows itself
        } else if (c->is2d()) {
            Probe2D q;
            q.g = grid2d(c);
            q.hx = c->h[0][c->hgen];
            q.hy = c->h[1][c->hgen];
            q.dz = c->D(2, 0);
            q.ez = c->E(2, 0);
            q.p = c->probes;
            q.ring = c->probe_ring;
            q.slot = slot;
            q.nmid = mid;
            for (int i = 0; i < mid; ++i) q.mid_slot[i] = (int)((c->step - 2 - i) % c->probe_cap);
            launch_probe2d(q, c->stream);
        } else {
            Probe3D q;
            q.g = grid3d(c);
            for (int k = 0; k < 3; ++k) {
                q.f[FDTD_HX + k] = c->h[k][c->hgen];
                q.f[FDTD_DX + k] = c->D(k, 0);
                q.f[FDTD_EX + k] = c->E(k, 0);
            }
            q.p = c->probes;
            q.ring = c->probe_ring;
            q.slot = slot;
            launch_probe3d(q, c->stream);
        }
        if (c->step - c->probe_first > c->probe_cap) c->probe_first = c->step - c->probe_cap;
    }
}

// Rows of a launch of the multi-step sweep over [y_begin, y_end): bands of tb_rows rows, the last
// eighth (at most 2048 rows) in bands a third as high (32 rows under tall bands)
// (16384^2, K = 8, 96 rows: no tail 3.63 ms, 1024 rows in 16s 3.50, 2048 in 32s 3.41).
void tb_rows_of(const fdtd_ctx *c, SweepTB &a, int y_begin, int y_end, int K)
{
    // A band on a y face runs the fully guarded copy of the loop for all its rows, which takes twice as
    // long per row as the plain one: with bands of the common height those waves set the length of a
    // launch of one or two rounds of waves (4096^2: 233 of 254 us).  The bands on the faces are as
    // short as the guards allow -- 2K+1 rows at y = 0, K+2 at y = ny-1 --, the bands next to them plain.
    static const bool short_faces = !std::getenv("FDTD_TB_SHORT_FACES") || std::atoi(std::getenv("FDTD_TB_SHORT_FACES"));
    a.head_rows = a.last_rows = 0;
    if (short_faces && y_end - y_begin >= 2 * c->tb_rows + 3 * K + 3) {
        if (c->s0 + y_begin == 0) a.head_rows = 2 * K + 1;
        if (c->s0 + y_end == c->gn) a.last_rows = K + 2;
    }
    const int yb = y_begin + a.head_rows;
    const int n = y_end - a.last_rows - yb;
    // (bands of 96 rows and more: 32-row tail bands -- 16384^2 at 176 rows: 790 against 770 with 58-row ones,
    // 16384 x 32768 796 against 789, 65536 x 4096 757 against 749, 10240^2 695 against 687, tissue map 680 against 685)
    int tail = std::min(2048, n / 8), short_rows = c->tb_rows >= 96 ? 32 : std::max(4, c->tb_rows / 3);
    if (const char *v = std::getenv("FDTD_TB_TAIL")) tail = std::atoi(v);
    if (const char *v = std::getenv("FDTD_TB_TAIL_ROWS")) short_rows = std::max(2, std::atoi(v));
    if (tail + c->tb_rows > n || short_rows >= c->tb_rows) tail = 0;
    a.rows = c->tb_rows;
    a.y_begin = y_begin;
    a.y_end = y_end;
    a.tail_rows = tail > 0 ? short_rows : c->tb_rows;
    a.big_bands = tail > 0 ? (n - tail) / c->tb_rows : (n + c->tb_rows - 1) / c->tb_rows;
    a.tail_begin = tail > 0 ? yb + a.big_bands * c->tb_rows : y_end - a.last_rows;
}

int finish_pass(fdtd_ctx *c, const SweepTB &a, int K, const Split &sp, int lo, int hi, bool tail);

// The waves of a launch (bands as tb_rows_of laid them out, map 0) that hold a source or a probe of
// this slab, for SweepTB::first.
void tb_first_waves(const fdtd_ctx *c, SweepTB &a, int K)
{
    a.nfirst = 0;
    a.faces_first = 0;
    if (a.map != 0) return;
    std::vector<std::pair<int, int>> bands; // rows [y0, y1) of every band, in launch order
    for (long b = 0, nb = tb_band_count(a); b < nb; ++b) {
        int y0 = 0, y1 = 0;
        if (tb_band_rows(a, b, y0, y1)) bands.push_back({y0, y1});
    }
    static const bool faces = !std::getenv("FDTD_TB_FACES_FIRST") || std::atoi(std::getenv("FDTD_TB_FACES_FIRST"));
    a.faces_first = (faces && a.strips > 2) ? (int)bands.size() : 0;
    std::vector<int> ids;
    auto add = [&](int x, int j, bool window) {
        // a source counts for every wave whose window (halo words and primed rows included) holds it,
        // a probe for the wave that owns its cell
        const int w = x >> 2;
        for (size_t b = 0; b < bands.size(); ++b) {
            const int lo = window ? bands[b].first - K : bands[b].first, hi = window ? bands[b].second + K - 2 : bands[b].second - 1;
            if (j < lo || j > hi) continue;
            // the strips whose words [w0, w1) hold word w: at most two
            const int s_lo = std::max(0, (w - kTbOwnWords - kTbHaloWords + 1) / kTbOwnWords - 1);
            for (int s = s_lo; s < std::min(a.strips, s_lo + 4); ++s) {
                const int w0 = s * kTbOwnWords - (window ? kTbHaloWords : 0), w1 = s * kTbOwnWords + kTbOwnWords + (window ? kTbHaloWords : 0);
                if (w >= w0 && w < w1) {
                    const int nb = a.faces_first;
                    // place of band b in the launch: the last band comes second
                    const int d = (!nb || nb < 3) ? (int)b : b == 0 ? 0 : (int)b == nb - 1 ? 1 : (int)b + 1;
                    ids.push_back(!nb ? (int)b * a.strips + s
                                      : s == 0 ? 2 * d : s == a.strips - 1 ? 2 * d + 1 : 2 * nb + d * (a.strips - 2) + s - 1);
                }
            }
        }
    };
    for (int s = 0; s < a.src.count; ++s) add(a.src.x[s], a.src.y[s] - c->s0, true);
    if (a.ring)
        for (int s = 0; s < a.probes.count; ++s)
            if (a.probes.x[s] >= 0 && a.probes.x[s] < c->nx) add(a.probes.x[s], a.probes.y[s] - c->s0, false);
    std::sort(ids.begin(), ids.end());
    ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    if (ids.size() > (size_t)kTbFirstWaves) ids.resize(kTbFirstWaves);
    a.nfirst = (int)ids.size();
    for (size_t i = 0; i < ids.size(); ++i) a.first[i] = ids[i];
}

// MULTISTEP: K <= tb_steps time steps in one pass (sweep2d_tb.hip), then Mur on the newest level;
// the tiled sweep (sweep2d_tile.hip) does all of it in one launch.
int pass_once(fdtd_ctx *c, int K)
{
    int rc = snapshot_guard(c);
    if (rc) return rc;
    const bool prof = c->prof_n < c->prof_cap;
    SweepTB a;
    a.g = grid2d(c);
    a.hx_r = c->h[0][c->hgen];
    a.hy_r = c->h[1][c->hgen];
    a.hx_w = c->h[0][c->hgen ^ 1];
    a.hy_w = c->h[1][c->hgen ^ 1];
    a.ez_n = c->E(2, 0);
    a.ez_p = c->E(2, 1);
    a.dz_n = c->D(2, 0);
    a.dz_p = c->D(2, 1);
    const int other = 2 * (c->pair ^ 1);
    a.ez_a = c->e[2][other];
    a.ez_b = c->e[2][other + 1];
    a.dz_a = c->d[2][other];
    a.dz_b = c->d[2][other + 1];
    a.code = c->code;
    a.table = c->table;
    a.table_wide = c->table_wide;
    a.one = c->table0;
    a.ntab = c->ntab_eff();
    a.src = c->src;
    a.src.jz = 0.0f;
    for (int s = 0; s < kMaxStepsPerPass; ++s) a.jz[s] = s < K ? jz_of_step(c, c->step + s) : 0.0f;
    a.rpx0 = c->rp[FDTD_RP_X0];
    a.rpxe = c->rp[FDTD_RP_XEND];
    a.cfx0 = c->mur_cf[FDTD_RP_X0];
    a.cfxe = c->mur_cf[FDTD_RP_XEND];
    a.cfy0 = c->mur_cf[FDTD_RP_Y0];
    a.cfye = c->mur_cf[FDTD_RP_YEND];
    a.rpy0 = c->rp[FDTD_RP_Y0];
    a.rpye = c->rp[FDTD_RP_YEND];
    a.dx = c->p.dx;
    a.dy = c->p.dy;
    a.mu0 = c->p.mu0;
    a.eps0 = c->p.eps0;
    a.probes = c->probes;
    a.ring = (c->probes.count > 0) ? c->probe_ring : nullptr;
    for (int s = 0; s < kMaxStepsPerPass; ++s)
        a.slot[s] = (a.ring && s >= 1 && s < K) ? (int)((c->step + s - 1) % c->probe_cap) : 0;
    a.strips = ((c->nx + 3) / 4 + kTbOwnWords - 1) / kTbOwnWords;
    a.map = c->tb_map;
    a.strips_per_xcd = (a.strips + 7) / 8;
    a.nfirst = a.faces_first = 0;
    a.head_rows = a.last_rows = 0;
    a.trace = nullptr;

    // The rows within `edge` of a neighbour slab go first, on xstream, and the deep-halo exchange
    // follows them there while the main stream sweeps the interior.
    const int edge = (c->tb_steps + 15) / 16 * 16;
    const Split sp = split_of(c, edge, c->tb_rows);
    const int lo = sp.on && sp.lower ? edge : 0, hi = sp.on && sp.upper ? c->sn - edge : c->sn;
    if (c->tile) {
        SweepTile t;
        t.b = a;
        t.steps = K;
        t.halo_rows = c->halo;
        t.slot_last = a.ring ? (int)((c->step + K - 1) % c->probe_cap) : 0;
        t.own_rows = t.tiles_x = 0;
        bool bad = false;
        if (sp.on) {
            if ((rc = fork_xstream(c))) return rc;
            const int ranges[2][2] = {{0, lo}, {hi, c->sn}};
            for (const auto &r : ranges) {
                if (r[1] <= r[0]) continue;
                t.b.y_begin = r[0];
                t.b.y_end = r[1];
                bad |= launch_sweep2d_tile(t, c->xstream) != 0;
            }
        }
        t.b.y_begin = lo;
        t.b.y_end = hi;
        if (prof) HIPCHK(hipEventRecord(c->prof_ev[2 * c->prof_n], c->stream));
        bad |= launch_sweep2d_tile(t, c->stream) != 0;
        if (prof) HIPCHK(hipEventRecord(c->prof_ev[2 * c->prof_n + 1], c->stream));
        if (bad) {
            set_error("the tiled sweep cannot run this pass (steps per pass outside 1 .. 8)");
            return -3;
        }
        return finish_pass(c, a, K, sp, lo, hi, false);
    }
    if (sp.on) {
        if ((rc = fork_xstream(c))) return rc;
        SweepTB e = a;
        e.rows = e.tail_rows = edge;
        e.head_rows = e.last_rows = 0;
        e.y_begin = 0;
        e.big_bands = lo > 0 ? 1 : 0;
        e.tail_begin = hi < c->sn ? hi : lo;
        e.y_end = hi < c->sn ? c->sn : lo;
        if (launch_sweep2d_tb(e, K, c->xstream)) {
            set_error("no multi-step kernel for this number of steps per pass");
            return -3;
        }
    }
    tb_rows_of(c, a, lo, hi, K);
    static const bool reorder = !std::getenv("FDTD_TB_FIRST") || std::atoi(std::getenv("FDTD_TB_FIRST"));
    if (reorder) tb_first_waves(c, a, K);
    if (c->wave_trace_armed) {
        // one launch: its positions are bands x strips, padded to whole blocks of at most 4 waves
        const long n = tb_band_count(a) * a.strips + 4;
        if (n <= c->wave_trace_cap) {
            HIPCHK(hipMemsetAsync(c->wave_trace, 0, (size_t)n * 32, c->stream));
            a.trace = c->wave_trace;
            c->wave_trace_n = n;
        }
        c->wave_trace_armed = false;
    }
    if (prof) HIPCHK(hipEventRecord(c->prof_ev[2 * c->prof_n], c->stream));
    if (launch_sweep2d_tb(a, K, c->stream)) {
        set_error("no multi-step kernel for this number of steps per pass");
        return -3;
    }
    if (prof) HIPCHK(hipEventRecord(c->prof_ev[2 * c->prof_n + 1], c->stream));
    return finish_pass(c, a, K, sp, lo, hi, true);
}

// What follows the sweep launches of a pass: the roles of the arrays swap, Mur on the newest level
// and the probe rows (`tail`: as launches of their own; the tiled sweep has done both itself), the
// halo exchange.
int finish_pass(fdtd_ctx *c, const SweepTB &a, int K, const Split &sp, int lo, int hi, bool tail)
{
    int rc;
    const bool prof = c->prof_n < c->prof_cap;
    c->pair ^= 1;
    c->cur = 1;
    c->hgen ^= 1;
    c->step += K;
    if (!tail) {
        HIPCHK(hipGetLastError());
        if (prof) c->prof_steps[c->prof_n++] = K;
        if (sp.on) {
            if ((rc = exchange_halos(c, c->xstream, false))) return rc;
            if ((rc = join_xstream(c))) return rc;
        } else if ((rc = exchange_halos(c, c->stream, false))) {
            return rc;
        }
        record_probes(c, K - 1, false);
        return 0;
    }

    Mur2D m = mur2d_args(c, a.g, c->E(2, 0), c->E(2, 1));
    if (sp.on) {
        const int ranges[2][2] = {{0, lo}, {hi, c->sn}};
        for (const auto &r : ranges) {
            if (r[1] <= r[0]) continue;
            m.j_begin = r[0];
            m.j_end = r[1];
            launch_mur2d(m, c->xstream);
        }
    }
    m.j_begin = lo;
    m.j_end = hi;
    launch_mur2d(m, c->stream);
    HIPCHK(hipGetLastError());
    if (prof) c->prof_steps[c->prof_n++] = K;

    if (sp.on) {
        if ((rc = exchange_halos(c, c->xstream, false))) return rc;
        if ((rc = join_xstream(c))) return rc;
    } else if ((rc = exchange_halos(c, c->stream, false))) {
        return rc;
    }
    record_probes(c, K - 1);
    return 0;
}

} // namespace

extern "C" {

void fdtd_options_default(fdtd_options *opt)
{
    std::memset(opt, 0, sizeof *opt);
    opt->mode = FDTD_MODE_AUTO;
    opt->ecoef = FDTD_ECOEF_REFERENCE;
    opt->kernel = FDTD_KERNEL_AUTO;
}

void fdtd_destroy(fdtd_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->side) (void)hipStreamSynchronize(c->side);
    if (c->xstream) (void)hipStreamSynchronize(c->xstream);
    if (c->comm) comm_destroy(c->comm);
    for (void *ptr : c->allocs) (void)hipFree(ptr);
    if (c->probe_ring) (void)hipFree(c->probe_ring);
    for (auto &sn : c->snap) {
        if (sn.dev) (void)hipFree(sn.dev);
        if (sn.host) (void)hipHostFree(sn.host);
        if (sn.copied) (void)hipEventDestroy(sn.copied);
        if (sn.done) (void)hipEventDestroy(sn.done);
    }
    if (c->snap_ev) (void)hipEventDestroy(c->snap_ev);
    if (c->io_stage) (void)hipFree(c->io_stage);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    for (hipEvent_t ev : c->prof_ev) (void)hipEventDestroy(ev);
    if (c->wave_trace) (void)hipFree(c->wave_trace);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->xstream) (void)hipStreamDestroy(c->xstream);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

static int create_impl(const fdtd_params *p, const fdtd_options *opt, fdtd_ctx *c)
{
    c->p = *p;
    c->nx = p->nx;
    c->ny = p->ny;
    c->nz = p->nz;
    c->mode = opt->mode == FDTD_MODE_AUTO ? (p->nz == 1 ? FDTD_MODE_2D_TMZ : FDTD_MODE_3D) : opt->mode;
    if (c->mode != FDTD_MODE_3D && c->mode != FDTD_MODE_2D_TMZ) {
        set_error("unknown mode");
        return -3;
    }
    if (c->is2d() && p->nz != 1) {
        set_error("the 2-D TMz mode needs nz = 1");
        return -3;
    }
    if (p->nx < 3 || p->ny < 3 || (!c->is2d() && p->nz < 3)) {
        set_error("every extent of the grid must be at least 3");
        return -3;
    }
    if (opt->ecoef != FDTD_ECOEF_REFERENCE && opt->ecoef != FDTD_ECOEF_CONSISTENT) {
        set_error("unknown ecoef");
        return -3;
    }
    c->ecoef = opt->ecoef;
    c->kernel = opt->kernel;
    c->auto_kernel = c->kernel == FDTD_KERNEL_AUTO;
    if (c->kernel == FDTD_KERNEL_AUTO) {
        // 2-D: the row-marching multi-step sweep from 7.5 M cells up; below, a pass of it is bound by
        // the latency of its march, not by bandwidth, and the tiled multi-step sweep is faster
        // (2048^2: 307 against 225 Gcell-updates/s, 1024^2: 211 against 81; 2560^2: 318 against 308; 2896^2:
        // 330 against 370, 3072^2: 315 against 406).
        // Decided from the extents of the WHOLE grid, so that every slab of it takes the same kernel.
        // Smaller 2-D grids are bound by launch and memory latency: the tiled sweep advances them
        // several steps per launch out of registers and LDS.
        c->kernel = !c->is2d() ? FDTD_KERNEL_FUSED : (double)p->nx * (double)p->ny >= 7.5e6 ? FDTD_KERNEL_MULTISTEP : FDTD_KERNEL_TILE;
    }
    if (c->kernel == FDTD_KERNEL_TILE) {
        // the storage, the schedule and the halo protocol of the multi-step mode, another sweep kernel
        c->tile = true;
        c->kernel = FDTD_KERNEL_MULTISTEP;
    }
    if (c->kernel != FDTD_KERNEL_SPLIT && c->kernel != FDTD_KERNEL_FUSED && c->kernel != FDTD_KERNEL_MULTISTEP) {
        set_error("unknown kernel selection");
        return -3;
    }
    if (c->kernel == FDTD_KERNEL_MULTISTEP) {
        if (!c->is2d()) {
            set_error("FDTD_KERNEL_MULTISTEP and FDTD_KERNEL_TILE exist for the 2-D TMz mode only");
            return -3;
        }
        // (steps per pass 6 / 7 / 8, tools/kscan.sh: 2896^2 365 / 370 / 352 Gcell-updates/s, 3072^2 395 / 406 /
        // 388, 3328^2 406 / 426 / 412, 3584^2 424 / 469 / 469, 3840^2 421 / 450 / 499);
        // from the extents of the whole grid: every slab must advance the same number of steps
        // between two halo exchanges
        const bool mid = (double)p->nx * (double)p->ny < 12e6;
        // (the tiled sweep from 3 M cells up, where its tiles take several rounds: 2048^2 329 at 6 steps per
        // launch, 312 at 8; 2896^2 347 against 333; 1024^2 205 against 210)
        const bool tile6 = c->tile && (double)p->nx * (double)p->ny >= 3e6;
        c->tb_steps = opt->steps_per_pass > 0 ? opt->steps_per_pass : (mid && !c->tile) ? 7 : tile6 ? 6 : kDefaultStepsPerPass;
        c->spp_fixed = opt->steps_per_pass > 0 || (opt->slab_count > 0 && opt->slab_count != p->ny);
        if (c->tb_steps < (c->tile ? 1 : 2) || c->tb_steps > kMaxStepsPerPass) {
            set_error("steps_per_pass must be 2 .. 8 (FDTD_KERNEL_TILE: 1 .. 8)");
            return -3;
        }
        c->tb_rows = 0; // chosen once the slab is known
        if (const char *v = std::getenv("FDTD_TB_ROWS")) c->tb_rows = std::max(2, std::atoi(v));
        c->halo = c->tb_steps;
    }
    c->gn = c->is2d() ? p->ny : p->nz;
    c->s0 = opt->slab_count > 0 ? opt->slab_begin : 0;
    c->sn = opt->slab_count > 0 ? opt->slab_count : c->gn;
    if (c->s0 < 0 || c->sn < 2 || c->s0 + c->sn > c->gn) {
        set_error("slab must hold at least 2 rows inside the grid");
        return -3;
    }
    if (c->kernel == FDTD_KERNEL_MULTISTEP && c->tb_rows == 0) {
        // Rows per wave, from the number of waves the GPU holds at once: 2048 (256 CUs x 2 workgroups of 4
        // waves: 244 registers and 20 KB of LDS per wave).  A wave marches its rows plus 2K - 1 around them
        // at about 3 us per row whatever the load (it is bound by the latency of its own dependency chain,
        // not by bandwidth, until some 2000 of them run), so a launch takes (rounds of waves) x (rows + 2K - 1)
        // x 3 us and wants few, tall waves -- but a launch of exactly one round ends when its slowest wave
        // ends (the strips on the x faces take 1.5 times as long as the others, a wave that holds a source
        // or a probe twice as long), and the short bands at the end of a launch only fill gaps if there
        // are gaps to fill.  Measured over 3584^2 ... 8192 x 4096 (tools/roundscan.sh): rows x strips /
        // (2048 x 1.45) is best or within 2 % of it (4096^2 435 Gcell-updates/s at 28 rows, 420 at 24, 419 at
        // 32; 5632^2 561 at 44, 530 at 52, 524 at 40; 7168^2 633 at 72, 596 at 80, 594 at 64); from 9216^2 up,
        // where 2048 waves saturate the memory system and the pass time follows the bytes, more and shorter
        // waves balance the end of the launch better: at most 176 rows (16384^2: 176 rows 798, 128 rows 777,
        // 256 rows 801), and below that just under two rounds (tools/roundscan2.sh, rounds 1.45 / 1.9 / 2.3 /
        // 2.9: 9216^2 685 / 699 / 689 / 688 Gcell-updates/s, 11264^2 684 / 735 / 710 / 712, 12288^2 724 / 746 /
        // 722 / 737, 8192 x 16384 781 / 808 / 788 / 786, the 8480 x 15680 tissue map 704 / 733 / 709 / 710).
        const long strips = ((c->nx + 3) / 4 + kTbOwnWords - 1) / kTbOwnWords;
        const long work = (long)c->sn * strips;
        double rounds = work < 60000 ? 1.2 : work < 330000 ? 1.45 : 1.9;
        if (const char *v = std::getenv("FDTD_TB_ROUNDS")) rounds = std::max(0.25, std::atof(v));
        const long want = (long)(work / (2048.0 * rounds) / 4.0 + 0.5) * 4;
        c->tb_rows = (int)std::min(176L, std::max(8L, want));
        if ((double)c->nx * c->sn < 6e6) c->tb_rows = (int)std::max(4L, std::min((long)c->tb_rows, (long)c->sn * strips / 2048));
    }
    if (c->kernel == FDTD_KERNEL_MULTISTEP && c->sn != c->gn && c->sn < c->tb_steps) {
        set_error("a slab of the multi-step mode must hold at least steps_per_pass rows");
        return -3;
    }
    c->chunk_rows = opt->chunk_rows > 0 ? opt->chunk_rows : 1;
    if (const char *v = std::getenv("FDTD_SWEEP_VARIANT")) c->variant = std::atoi(v);
    if (const char *v = std::getenv("FDTD_SWEEP_BAND")) c->band = std::atoi(v) / 8 * 8;
    if (const char *v = std::getenv("FDTD_SWEEP_NOPAD")) c->nopad = std::atoi(v);
    if (p->sources_count < 0 || p->sources_count > FDTD_MAX_SOURCES) {
        set_error("too many sources");
        return -3;
    }
    c->src.count = p->sources_count;
    c->src.jz = 0.0f;
    for (int i = 0; i < p->sources_count; ++i) {
        const int sx = p->sources[i * 3], sy = p->sources[i * 3 + 1], sz = c->is2d() ? 0 : p->sources[i * 3 + 2];
        // updateSources reads hy[x-1] and hx[y-1] unguarded (fdtd_calculations.cu:206-207)
        if (sx < 1 || sx > p->nx - 1 || sy < 1 || sy > p->ny - 1 || sz < 0 || sz > p->nz - 1) {
            set_error("source_location outside 2..nx, 2..ny, 1..nz");
            return -3;
        }
        c->src.x[i] = sx;
        c->src.y[i] = sy;
        c->src.z[i] = sz;
    }

    c->device = opt->device;
    HIPCHK(hipSetDevice(c->device));
    (void)scratch_init(c->device);
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    {
        int least = 0, greatest = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        if (std::getenv("FDTD_XSTREAM_NORMAL")) greatest = 0;
        HIPCHK(hipStreamCreateWithPriority(&c->xstream, hipStreamNonBlocking, greatest));
    }
    HIPCHK(hipEventCreateWithFlags(&c->snap_ev, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    if (const char *v = std::getenv("FDTD_HALO_OVERLAP")) c->overlap = std::atoi(v) != 0;
    if (const char *v = std::getenv("FDTD_TB_MAP")) c->tb_map = std::atoi(v);

    int skew = 0;
    if (const char *s = std::getenv("FDTD_PITCH_SKEW")) skew = std::atoi(s);
    // 2-D: rows start on a 256-byte line (a wave owns 256 columns of one row).  3-D fused: a
    // plane is one flat run of 16-byte words, so rows only need 64-byte alignment and a narrow
    // grid (nx = 265) wastes 3 % of the lanes instead of 17 %.
    // Small 3-D grids (up to kFlat3dCells per slab): rows on 16-byte boundaries only, and a launch
    // takes all its planes as one run of words (100^3: 3900 waves instead of 4800, 31.0 -> 29.0 us/step).
    // Taken where it saves a tenth of the waves; planes that fill their segments anyway keep them (an XCD
    // then re-reads the planes z-1 and z+1 from its own L2: 265 x 490 x 8 is 13 % faster that way).
    c->flat3d = false;
    if (!c->is2d() && c->kernel == FDTD_KERNEL_FUSED && (long)c->nx * c->ny * c->sn <= kFlat3dCells) {
        auto waves_of = [](long words) { return ((words + 7) / 8 + 63) / 64 * 8; };
        const long per_plane = waves_of((long)c->ny * ((c->nx + 15) / 16 * 4)) * c->sn;
        const long flat = waves_of((long)c->sn * c->ny * ((c->nx + 3) / 4));
        c->flat3d = flat * 10 <= per_plane * 9;
    }
    if (const char *v = std::getenv("FDTD_3D_FLAT")) c->flat3d = c->flat3d && std::atoi(v) != 0;
    const int quantum = (!c->is2d() && c->kernel == FDTD_KERNEL_FUSED) ? (c->flat3d ? 4 : 16) : 64;
    const int row_pitch = (c->nx + quantum - 1) / quantum * quantum + skew * 64;
    // Multi-step sweep: the six arrays a pass reads (H and both levels of ez and dz) are interleaved
    // row by row in one allocation, and so are the six it writes: a band of waves marching up the
    // grid then walks ONE run of memory per direction, six rows at a time, instead of six runs far
    // apart, which is what DRAM pages want (16384^2: FDTD_TB_INTERLEAVE=0 gives the separate arrays).
    bool interleave = c->kernel == FDTD_KERNEL_MULTISTEP;
    if (const char *v = std::getenv("FDTD_TB_INTERLEAVE")) interleave = interleave && std::atoi(v) != 0;
    // Fused 3-D sweep: the three components of a field are interleaved row by row (row y of ex, of ey,
    // of ez, then row y + 1 ...): 8 runs of memory per sweep instead of 24 (FDTD_3D_INTERLEAVE=0: apart).
    c->interleave3 = !c->is2d() && c->kernel == FDTD_KERNEL_FUSED;
    if (const char *v = std::getenv("FDTD_3D_INTERLEAVE")) c->interleave3 = c->interleave3 && std::atoi(v) != 0;
    // Small 3-D grids: a step is the sweep and the Mur launch, and the Mur launch -- 5 us whatever the
    // grid -- is a sixth of a 100^3 step; the sweep finishes the faces itself there (two waves per
    // SIMD, which costs a large grid more than the launch it saves).  FDTD_3D_FOLD_MUR=0/1 overrides.
    c->fold_mur = !c->is2d() && c->kernel == FDTD_KERNEL_FUSED && c->sn == c->gn && c->nx >= 4 && c->ny >= 4 &&
                  c->nz >= 4 && (long)c->nx * c->ny * c->nz <= kFoldMurCells;
    if (const char *v = std::getenv("FDTD_3D_FOLD_MUR"))
        c->fold_mur = std::atoi(v) != 0 && !c->is2d() && c->kernel == FDTD_KERNEL_FUSED && c->sn == c->gn &&
                      c->nx >= 4 && c->ny >= 4 && c->nz >= 4;
    c->pitch = interleave ? 6 * row_pitch : c->interleave3 ? 3 * row_pitch : row_pitch;
    c->cpitch = row_pitch;
    c->rows_store = c->is2d() ? (size_t)c->sn + 2 * c->halo : ((size_t)c->sn + 2) * c->ny;
    c->elems = c->rows_store * c->pitch;
    c->lead = (size_t)(c->halo - 1) * c->pitch;
    c->celems = c->rows_store * (size_t)c->cpitch;
    c->clead = (size_t)(c->halo - 1) * c->cpitch;

    int rc = 0;
    if (interleave) {
        for (int grp = 0; grp < 2 && !rc; ++grp) {
            float *base = nullptr;
            if ((rc = dev_alloc(c, &base, c->elems + 2 * kSlack))) break;
            float *first = base + kSlack + c->lead;
            c->state_bufs.push_back({base + kSlack, c->elems});
            c->h[0][grp] = first;
            c->h[1][grp] = first + row_pitch;
            c->e[2][2 * grp] = first + 2 * (size_t)row_pitch;
            c->e[2][2 * grp + 1] = first + 3 * (size_t)row_pitch;
            c->d[2][2 * grp] = first + 4 * (size_t)row_pitch;
            c->d[2][2 * grp + 1] = first + 5 * (size_t)row_pitch;
        }
    } else if (c->interleave3) {
        // groups of three: H (two generations), E and D (two levels each)
        float **groups[6] = {&c->h[0][0], &c->h[0][1], &c->e[0][0], &c->e[0][1], &c->d[0][0], &c->d[0][1]};
        const size_t comp_step[6] = {2, 2, 4, 4, 4, 4}; // elements between [k][i] and [k + 1][i] of h[3][2], e[3][4], d[3][4]
        for (int grp = 0; grp < 6 && !rc; ++grp) {
            float *base = nullptr;
            if ((rc = dev_alloc(c, &base, c->elems + 2 * kSlack))) break;
            c->state_bufs.push_back({base + kSlack, c->elems});
            for (int k = 0; k < 3; ++k) groups[grp][k * comp_step[grp]] = base + kSlack + c->lead + (size_t)k * row_pitch;
        }
    } else {
        const bool two_h = c->one_pass();
        const int levels = c->kernel == FDTD_KERNEL_MULTISTEP ? 4 : 2;
        for (int k = 0; k < 3 && !rc; ++k) {
            const bool used_h = c->is2d() ? k < 2 : true;  // hx, hy
            const bool used_ed = c->is2d() ? k == 2 : true; // dz, ez
            if (used_h) {
                rc = alloc_field(c, &c->h[k][0]);
                if (!rc && two_h) rc = alloc_field(c, &c->h[k][1]);
            }
            for (int l = 0; l < levels && used_ed && !rc; ++l) {
                rc = alloc_field(c, &c->e[k][l]);
                if (!rc) rc = alloc_field(c, &c->d[k][l]);
            }
        }
    }
    if (rc) return rc;
    uint8_t *code_base = nullptr;
    if ((rc = dev_alloc(c, &code_base, c->celems + 2 * kSlack))) return rc;
    c->code = code_base + kSlack + c->clead;
    if ((rc = dev_alloc(c, &c->table, (size_t)FDTD_MAX_CODES))) return rc;
    if ((rc = dev_alloc(c, &c->table_wide, (size_t)FDTD_MAX_CODES))) return rc;

    // Every code maps to the defaults until a spec table is loaded.
    const MaterialRow def = {p->default_sigma, p->default_eps_s, p->default_eps_i, p->default_tau_d};
    for (int i = 0; i < FDTD_MAX_CODES; ++i) c->mats[i] = def;
    if ((rc = upload_table(c))) return rc;

    // setupMurBoundary on the defaults (fdtd.cu:59 runs before :65).  Of every plane the context holds
    // the run its kernels read: a 2-D slab the x-face entries of its rows and of the rows around them
    // that a multi-step pass looks at (a wave of the row-marching sweep primes its pipeline up to
    // 2 * steps_per_pass rows below its band; a tile of the tiled sweep is up to 64 rows tall whatever
    // the slab holds), both y-face rows, nothing of the z faces; a 3-D slab the x- and y-face entries
    // of its planes and both z-face planes.
    const size_t gy = (size_t)p->ny, gz = (size_t)p->nz, gx = (size_t)p->nx;
    c->rp_gn[FDTD_RP_X0] = c->rp_gn[FDTD_RP_XEND] = 2 * gy * gz;
    c->rp_gn[FDTD_RP_Y0] = c->rp_gn[FDTD_RP_YEND] = gx * 2 * gz;
    c->rp_gn[FDTD_RP_Z0] = c->rp_gn[FDTD_RP_ZEND] = gx * gy * 2;
    const float rx = rp_value(*p, def, true), ryz = rp_value(*p, def, false);
    for (int f = 0; f < 6; ++f) {
        const int axis = f % 3;
        size_t first = 0, n = c->rp_gn[f];
        if (c->is2d()) {
            if (axis == 0) {
                const long margin = 64 + 2 * kMaxStepsPerPass + 16;
                const long lo = std::max(0L, (long)c->s0 - margin), hi = std::min((long)c->gn, (long)c->s0 + c->sn + margin);
                first = 2 * (size_t)lo;
                n = 2 * (size_t)(hi - lo);
            } else if (axis == 2) {
                n = 0;
            }
        } else if (axis != 2) {
            const size_t per_plane = axis == 0 ? 2 * gy : 2 * gx;
            first = (size_t)c->s0 * per_plane;
            n = (size_t)c->sn * per_plane;
        }
        c->rp_first[f] = first;
        c->rp_n[f] = n;
        c->rp_default[f] = axis == 0 ? rx : ryz;
        c->rp_host[f].assign(n, c->rp_default[f]);
        if (n == 0) continue;
        if ((rc = dev_alloc(c, &c->rp_store[f], n))) return rc;
        c->rp[f] = c->rp_store[f] - first;
        if ((rc = upload_rp(c, f))) return rc;
    }
    if (c->deep() && !c->tile) preload_sweep2d_tb(c->stream);
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int fdtd_create(const fdtd_params *p, const fdtd_options *opt, fdtd_ctx **out)
{
    if (!p || !out) {
        set_error("fdtd_create: null argument");
        return -3;
    }
    fdtd_options o;
    if (opt)
        o = *opt;
    else
        fdtd_options_default(&o);
    fdtd_ctx *c = new fdtd_ctx();
    int rc = create_impl(p, &o, c);
    if (rc == -10 && c->auto_kernel && c->kernel == FDTD_KERNEL_MULTISTEP) {
        // FDTD_KERNEL_AUTO took the multi-step sweep and its 49 bytes per cell did not fit.  A whole
        // grid falls back to the one-step sweep (33 bytes per cell); a slab must not decide on its
        // own, because its neighbours would keep another halo depth.
        const bool whole = o.slab_count <= 0 || o.slab_count == (p->nz == 1 ? p->ny : p->nz);
        const std::string keep = fdtd_last_error();
        fdtd_destroy(c);
        (void)hipGetLastError();
        if (!whole) {
            set_error(keep + " (FDTD_KERNEL_AUTO chose the multi-step sweep from the extents of the whole grid and "
                             "this slab's arrays do not fit: pass FDTD_KERNEL_FUSED to every rank)");
            return rc;
        }
        o.kernel = FDTD_KERNEL_FUSED;
        c = new fdtd_ctx();
        rc = create_impl(p, &o, c);
    }
    if (rc) {
        const std::string keep = fdtd_last_error();
        fdtd_destroy(c);
        set_error(keep);
        return rc;
    }
    *out = c;
    return 0;
}

int fdtd_set_material(fdtd_ctx *c, int code, float sigma, float eps_s, float eps_inf, float tau_d)
{
    if (code < 0 || code >= FDTD_MAX_CODES) {
        set_error("material code out of range 0..255");
        return -3;
    }
    if (use_device(c)) return -10;
    c->mats[code] = {sigma, eps_s, eps_inf, tau_d};
    return upload_table(c);
}

int fdtd_set_codes(fdtd_ctx *c, const uint8_t *codes, size_t count)
{
    if (count != c->slab_cells()) {
        set_error("fdtd_set_codes: count must be the number of cells of the slab");
        return -3;
    }
    if (use_device(c)) return -10;
    c->codes_host.assign(codes, codes + count);
    c->codes_up = 0;
    c->codes_base.clear();
    c->halo_stale = true;
    return upload_codes(c);
}

int fdtd_load_materials(fdtd_ctx *c, const char *specs_path, const char *pgm_dir, int first_slice, int upsample)
{
    if (use_device(c)) return -10;
    bool present[FDTD_MAX_CODES] = {};
    MaterialRow table[FDTD_MAX_CODES];
    // codes without a row in the spec file get zeros, as calloc gives them (fdtd.cu:879)
    std::memset(table, 0, sizeof table);
    int rc = read_material_specs(specs_path, table, present);
    if (rc) return rc;
    const MaterialRow def = c->mats[kDefaultCode];
    for (int i = 0; i < FDTD_MAX_CODES; ++i) c->mats[i] = table[i];
    c->mats[kDefaultCode] = def;

    c->codes_host.clear();
    c->codes_base.clear();
    c->codes_up = 0;
    const int planes = c->is2d() ? 1 : c->sn;
    if (upsample > 1) {
        // The slice images go to the device as they are and a kernel repeats every pixel
        // upsample x upsample times into the code array: a 65536^2 map never exists on the host.
        if (upsample < 1 || c->nx % upsample || c->ny % upsample) {
            set_error("upsample must divide nx and ny");
            return -3;
        }
        const int bx = c->nx / upsample, by = c->ny / upsample;
        c->codes_base.resize((size_t)planes * bx * by);
        for (int k = 0; k < planes; ++k) {
            char path[2048];
            std::snprintf(path, sizeof path, "%s/v1_%05d.pgm", pgm_dir, (c->is2d() ? 0 : c->s0 + k) + first_slice);
            if ((rc = read_pgm_base(path, bx, by, c->codes_base.data() + (size_t)k * bx * by))) return rc;
        }
        c->codes_up = upsample;
        c->base_x = bx;
        c->base_y = by;
        // device codes; the highest one among the pixels this slab's rows come from
        std::vector<uint8_t> dev(c->codes_base.size());
        const int r0 = c->is2d() ? c->s0 / upsample : 0, r1 = c->is2d() ? (c->s0 + c->sn - 1) / upsample : by - 1;
        int top = 0;
        for (int k = 0; k < planes; ++k)
            for (int r = 0; r < by; ++r)
                for (int x = 0; x < bx; ++x) {
                    const size_t i = ((size_t)k * by + r) * bx + x;
                    dev[i] = device_code(c->codes_base[i]);
                    if (r >= r0 && r <= r1) top = std::max(top, (int)dev[i]);
                }
        bool same = true;
        if (!c->is2d() || (r0 == 0 && r1 == by - 1)) {
            for (size_t i = 1; i < dev.size() && same; ++i) same = dev[i] == dev[0];
        } else {
            same = false; // a slab of an upsampled map: only the highest code of its rows is looked at
        }
        uint8_t *dbase = nullptr;
        HIPCHK(hipMalloc((void **)&dbase, dev.size()));
        hipError_t e = hipMemcpyAsync(dbase, dev.data(), dev.size(), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) {
            ExpandCodes x;
            x.code = c->code + (c->is2d() ? (size_t)c->cpitch : (size_t)c->ny * c->cpitch);
            x.base = dbase;
            x.nx = c->nx;
            x.bx = bx;
            x.by = by;
            x.up = upsample;
            x.rows = c->is2d() ? c->sn : c->ny;
            x.row0 = c->is2d() ? c->s0 : 0;
            x.planes = planes;
            x.row_pitch = c->cpitch;
            x.plane_pitch = c->is2d() ? 0 : (long long)c->ny * c->cpitch;
            launch_expand_codes(x, c->stream);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        (void)hipFree(dbase);
        if (e != hipSuccess) {
            set_error(std::string("fdtd_load_materials: ") + hipGetErrorString(e));
            return -10;
        }
        if ((rc = set_code_range(c, top, same && !dev.empty() ? (int)dev[0] : -1))) return rc;
        c->halo_stale = true;
        return 0;
    }
    c->codes_host.assign(c->slab_cells(), 0);
    for (int k = 0; k < planes; ++k) {
        char path[2048];
        std::snprintf(path, sizeof path, "%s/v1_%05d.pgm", pgm_dir, (c->is2d() ? 0 : c->s0 + k) + first_slice);
        rc = c->is2d() ? read_pgm_slice(path, c->nx, c->ny, 1, c->s0, c->sn, c->codes_host.data())
                       : read_pgm_slice(path, c->nx, c->ny, 1, 0, c->ny, c->codes_host.data() + (size_t)k * c->nx * c->ny);
        if (rc) return rc;
    }
    if ((rc = upload_table(c))) return rc;
    c->halo_stale = true;
    return upload_codes(c);
}

int fdtd_set_rp(fdtd_ctx *c, int face, const float *data, size_t count)
{
    if (face < 0 || face > 5 || count != c->rp_gn[face] || !data) {
        set_error("fdtd_set_rp: bad face, or count is not the size of the whole plane");
        return -3;
    }
    if (use_device(c)) return -10;
    // the context keeps the entries its kernels read (all of them on a whole 3-D grid)
    const float *run = data + c->rp_first[face];
    c->rp_host[face].assign(run, run + c->rp_n[face]);
    if (face == FDTD_RP_X0 || face == FDTD_RP_XEND) c->rp_from_codes = false; // the caller's plane stands
    return upload_rp(c, face);
}

int fdtd_get_rp(fdtd_ctx *c, int face, float *data, size_t count)
{
    if (face < 0 || face > 5 || count != c->rp_gn[face] || !data) {
        set_error("fdtd_get_rp: bad face, or count is not the size of the whole plane");
        return -3;
    }
    // entries the context does not hold read as the values derived from the defaults
    std::fill(data, data + c->rp_first[face], c->rp_default[face]);
    std::memcpy(data + c->rp_first[face], c->rp_host[face].data(), c->rp_n[face] * sizeof(float));
    std::fill(data + c->rp_first[face] + c->rp_n[face], data + count, c->rp_default[face]);
    return 0;
}

int fdtd_setup_mur_from_materials(fdtd_ctx *c)
{
    if (use_device(c)) return -10;
    if (c->deep() && c->sn != c->gn) {
        // a multi-step slab also applies the Mur update to the neighbours' rows it recomputes in its
        // halos: their x-face entries are derived from the material codes that arrive with the next
        // exchange, which this call makes due
        c->rp_from_codes = true;
        c->halo_stale = true;
    }
    const int nx = c->nx, ny = c->ny, nz = c->nz;
    // A slab fills in the entries of the cells it holds -- the only ones its Mur launches look at;
    // the rest of its planes keeps the values derived from the defaults.
    auto held = [&](int y, int z) { const int s = c->is2d() ? y : z; return s >= c->s0 && s < c->s0 + c->sn; };
    auto mat = [&](int x, int y, int z) -> const MaterialRow & {
        const int ly = c->is2d() ? y - c->s0 : y, lz = c->is2d() ? 0 : z - c->s0;
        if (c->codes_up > 0)
            return c->mats[c->codes_base[((size_t)lz * c->base_y + y / c->codes_up) * c->base_x + x / c->codes_up]];
        if (c->codes_host.empty()) return c->mats[kDefaultCode];
        return c->mats[c->codes_host[((size_t)lz * (c->is2d() ? c->sn : ny) + ly) * nx + x]];
    };
    // entry r of the whole plane of `face` (held cells lie inside the run the context keeps)
    auto at = [&](int face, size_t r) -> float & { return c->rp_host[face][r - c->rp_first[face]]; };
    const int z_lo = c->is2d() ? 0 : c->s0, z_hi = c->is2d() ? 1 : c->s0 + c->sn;
    const int y_lo = c->is2d() ? c->s0 : 0, y_hi = c->is2d() ? c->s0 + c->sn : ny;
    for (int z = z_lo; z < z_hi; ++z)
        for (int y = y_lo; y < y_hi; ++y)
            for (int k = 0; k < 2; ++k) {
                const size_t r = ((size_t)z * ny + y) * 2 + k;
                at(FDTD_RP_X0, r) = rp_value(c->p, mat(k, y, z), true);
                at(FDTD_RP_XEND, r) = rp_value(c->p, mat(nx - 2 + k, y, z), true);
            }
    for (int z = z_lo; z < z_hi; ++z)
        for (int k = 0; k < 2; ++k)
            for (int x = 0; x < nx; ++x) {
                const size_t r = ((size_t)z * 2 + k) * nx + x;
                if (held(k, z)) at(FDTD_RP_Y0, r) = rp_value(c->p, mat(x, k, z), false);
                if (held(ny - 2 + k, z)) at(FDTD_RP_YEND, r) = rp_value(c->p, mat(x, ny - 2 + k, z), false);
            }
    if (!c->is2d())
        for (int k = 0; k < 2; ++k)
            for (int y = 0; y < ny; ++y)
                for (int x = 0; x < nx; ++x) {
                    const size_t r = ((size_t)k * ny + y) * nx + x;
                    if (held(y, k)) at(FDTD_RP_Z0, r) = rp_value(c->p, mat(x, y, k), false);
                    if (held(y, nz - 2 + k)) at(FDTD_RP_ZEND, r) = rp_value(c->p, mat(x, y, nz - 2 + k), false);
                }
    for (int f = 0; f < 6; ++f) {
        int rc = upload_rp(c, f);
        if (rc) return rc;
    }
    return 0;
}

int fdtd_step(fdtd_ctx *c, int nsteps)
{
    if (use_device(c)) return -10;
    const int spp = fdtd_steps_per_pass(c);
    if (c->deep() && c->sn != c->gn) {
        if (c->comm && c->world > 1) {
            if (c->halo_stale) {
                // codes, fields or a checkpoint were loaded since the last exchange: the deep halos
                // (which hold the neighbours' materials and previous levels too) are refreshed first
                int rc = exchange_halos(c, c->stream, true);
                if (rc) return rc;
            }
        } else if (nsteps > spp) {
            set_error("fdtd_step: a slab of the multi-step mode without a communicator advances at most "
                      "steps_per_pass steps per call; exchange the halos (fdtd_halo_get/put) between calls");
            return -3;
        }
    }
    for (int i = 0; i < nsteps;) {
        // full passes, then one shorter pass for the remainder (a single step: the one-step sweep)
        const int k = std::min(nsteps - i, spp);
        int rc = (k >= 2 || c->tile) ? pass_once(c, k) : step_once(c, i + k >= nsteps);
        if (rc) return rc;
        i += k;
    }
    return 0;
}

int fdtd_steps_per_pass(const fdtd_ctx *c)
{
    if (c->kernel != FDTD_KERNEL_MULTISTEP) return 1;
    // Several materials on a whole grid: up to kWideTableRowsK8 table rows fit beside the parked rows
    // of 8 steps per pass; beyond, 7 steps per pass with four waves per workgroup beat 8 with three.
    // Slabs keep the number they were created with: every rank must advance the same number of steps
    // between two exchanges whatever materials it holds.
    if (c->spp_fixed || c->tile || c->ntab_eff() <= kWideTableRowsK8) return c->tb_steps;
    return std::min(c->tb_steps, kFastStepsPerPassTable);
}

int fdtd_sync(fdtd_ctx *c)
{
    if (use_device(c)) return -10;
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int64_t fdtd_steps_done(const fdtd_ctx *c) { return c->step; }

int fdtd_profile_begin(fdtd_ctx *c, int capacity)
{
    if (use_device(c)) return -10;
    HIPCHK(hipStreamSynchronize(c->stream));
    while ((int)c->prof_ev.size() < 2 * capacity) {
        hipEvent_t ev;
        HIPCHK(hipEventCreate(&ev));
        c->prof_ev.push_back(ev);
    }
    c->prof_steps.assign((size_t)std::max(capacity, 0), 0);
    c->prof_cap = capacity;
    c->prof_n = 0;
    return 0;
}

int fdtd_profile_read_launches(fdtd_ctx *c, float *sweep_ms, int32_t *steps, int capacity, int *count)
{
    if (use_device(c)) return -10;
    HIPCHK(hipStreamSynchronize(c->stream));
    const int n = std::min(capacity, c->prof_n);
    for (int i = 0; i < n; ++i) {
        HIPCHK(hipEventElapsedTime(&sweep_ms[i], c->prof_ev[2 * i], c->prof_ev[2 * i + 1]));
        if (steps) steps[i] = c->prof_steps[i];
    }
    if (count) *count = n;
    c->prof_cap = 0;
    return 0;
}

int fdtd_profile_waves_begin(fdtd_ctx *c, int64_t capacity)
{
    if (use_device(c)) return -10;
    if (capacity > c->wave_trace_cap) {
        if (c->wave_trace) (void)hipFree(c->wave_trace);
        c->wave_trace = nullptr;
        c->wave_trace_cap = 0;
        HIPCHK(hipMalloc(&c->wave_trace, (size_t)capacity * 32));
        c->wave_trace_cap = capacity;
    }
    c->wave_trace_n = 0;
    c->wave_trace_armed = capacity > 0;
    return 0;
}

int fdtd_profile_waves_read(fdtd_ctx *c, uint64_t *dst, int64_t capacity, int64_t *count)
{
    if (use_device(c)) return -10;
    HIPCHK(hipStreamSynchronize(c->stream));
    const long n = std::min<long>(capacity, c->wave_trace_n);
    if (n > 0) HIPCHK(hipMemcpy(dst, c->wave_trace, (size_t)n * 32, hipMemcpyDeviceToHost));
    if (count) *count = n;
    return 0;
}

int fdtd_profile_read(fdtd_ctx *c, float *sweep_ms, int capacity, int *count)
{
    return fdtd_profile_read_launches(c, sweep_ms, nullptr, capacity, count);
}

int fdtd_sweep_work(const fdtd_ctx *c, int64_t *waves, int64_t *row_iterations)
{
    if (c->tile) {
        // one workgroup per tile of 48 x (tile rows - 2 * steps) owned cells, one wave per 4 rows
        // (two words per thread: per 8), one iteration per step; a slab whose edge rows go in launches
        // of their own (halo exchange beside the interior sweep) is counted launch by launch
        const int K = c->tb_steps;
        const int edge = (c->tb_steps + 15) / 16 * 16;
        const Split sp = split_of(c, edge, c->tb_rows);
        const int lo = sp.on && sp.lower ? edge : 0, hi = sp.on && sp.upper ? c->sn - edge : c->sn;
        const int ranges[3][2] = {{0, lo}, {lo, hi}, {hi, c->sn}};
        int64_t w = 0;
        for (const auto &r : ranges) {
            if (r[1] <= r[0]) continue;
            const int rows = sweep2d_tile_rows(c->nx, r[1] - r[0], K);
            const int ty = rows == 642 ? 64 : rows, waves_per_tile = rows == 642 ? 8 : rows / 4;
            if (ty - 2 * K < 1) continue;
            const int64_t tiles = (int64_t)(((c->nx + 3) / 4 + 11) / 12) * ((r[1] - r[0] + ty - 2 * K - 1) / (ty - 2 * K));
            w += tiles * waves_per_tile;
        }
        if (waves) *waves = w;
        if (row_iterations) *row_iterations = w * K;
        return 0;
    }
    if (!c->deep()) {
        // one-step kernels: a wave handles one row (2-D: of 256 columns; 3-D: 64 words of a plane) once
        const int64_t w = c->is2d() ? (int64_t)((c->nx + 255) / 256) * c->sn
                                    : ((int64_t)c->ny * (c->cpitch / 4) + 63) / 64 * c->sn;
        if (waves) *waves = w;
        if (row_iterations) *row_iterations = w;
        return 0;
    }
    SweepTB a;
    const int K = fdtd_steps_per_pass(c);
    tb_rows_of(c, a, 0, c->sn, K);
    const int64_t strips = ((c->nx + 3) / 4 + kTbOwnWords - 1) / kTbOwnWords;
    const int64_t bands = tb_band_count(a);
    int64_t iters = 0;
    for (int64_t b = 0; b < bands; ++b) {
        int y0 = 0, y1 = 0;
        if (!tb_band_rows(a, b, y0, y1)) continue;
        // (the bands away from the y faces fill their pipeline with K(K-1) stage executions less: the
        // worth of K - 1 iterations; the head and the last band run the fully guarded loop, which does not)
        const bool guarded = (a.head_rows > 0 && b == 0) || (a.last_rows > 0 && b == bands - 1);
        iters += y1 - y0 + (guarded ? 2 * K - 1 : K);
    }
    if (waves) *waves = bands * strips;
    if (row_iterations) *row_iterations = iters * strips;
    return 0;
}

const char *fdtd_sweep_kernel_name(const fdtd_ctx *c)
{
    if (c->is2d() && c->kernel == FDTD_KERNEL_MULTISTEP) return c->tile ? sweep2d_tile_kernel_name() : sweep2d_tb_kernel_name();
    if (c->is2d()) return c->kernel == FDTD_KERNEL_FUSED ? sweep2d_kernel_name() : de2d_kernel_name();
    return c->kernel == FDTD_KERNEL_FUSED ? sweep3d_kernel_name() : de3d_kernel_name();
}

// Rows of a field between the host (rows of ncols floats, contiguous) and the device (rows `pitch`
// floats apart).  A 2-D copy between pageable host memory and the device goes row by row: fine for the
// 64 KB rows of a large 2-D grid, 130 MB/s for the 1 KB rows of a 265-cell-wide 3-D grid (7 s per
// field of the body model).  Short rows are gathered into / scattered from a contiguous staging buffer
// on the device -- a device-side 2-D copy at memory speed -- and cross PCIe as one block per chunk.
static int copy_rows(fdtd_ctx *c, float *host, float *dev, size_t rows, size_t ncols, bool to_host)
{
    const size_t row_bytes = ncols * sizeof(float);
    if (row_bytes >= 32768 || rows < 256) {
        if (to_host)
            HIPCHK(hipMemcpy2DAsync(host, row_bytes, dev, (size_t)c->pitch * 4, row_bytes, rows, hipMemcpyDeviceToHost, c->stream));
        else
            HIPCHK(hipMemcpy2DAsync(dev, (size_t)c->pitch * 4, host, row_bytes, row_bytes, rows, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        return 0;
    }
    const size_t want = std::min(rows * ncols, (size_t)64 << 20); // at most 256 MB of staging
    if (c->io_stage_floats < want) {
        if (c->io_stage) (void)hipFree(c->io_stage);
        c->io_stage = nullptr;
        c->io_stage_floats = 0;
        HIPCHK(hipMalloc((void **)&c->io_stage, want * sizeof(float)));
        c->io_stage_floats = want;
    }
    const size_t chunk = std::max<size_t>(1, c->io_stage_floats / ncols);
    for (size_t r = 0; r < rows; r += chunk) {
        const size_t n = std::min(chunk, rows - r);
        float *d = dev + r * (size_t)c->pitch, *h = host + r * ncols;
        if (to_host) {
            HIPCHK(hipMemcpy2DAsync(c->io_stage, row_bytes, d, (size_t)c->pitch * 4, row_bytes, n, hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(hipMemcpyAsync(h, c->io_stage, n * row_bytes, hipMemcpyDeviceToHost, c->stream));
        } else {
            HIPCHK(hipMemcpyAsync(c->io_stage, h, n * row_bytes, hipMemcpyHostToDevice, c->stream));
            HIPCHK(hipMemcpy2DAsync(d, (size_t)c->pitch * 4, c->io_stage, row_bytes, row_bytes, n, hipMemcpyDeviceToDevice, c->stream));
        }
        HIPCHK(hipStreamSynchronize(c->stream)); // the staging buffer is reused by the next chunk
    }
    return 0;
}

int fdtd_read_field(fdtd_ctx *c, int field, int level, float *dst, size_t count)
{
    if (use_device(c)) return -10;
    FieldView v;
    int rc = field_view(c, field, level, &v);
    if (rc) return rc;
    if (count != v.rows * c->nx) {
        set_error("fdtd_read_field: count must be the number of cells of the slab");
        return -3;
    }
    return copy_rows(c, dst, v.ptr, v.rows, (size_t)c->nx, true);
}

int fdtd_read_window(fdtd_ctx *c, int field, int level, int64_t first_row, int64_t nrows, int x0, int ncols,
                     float *dst)
{
    if (use_device(c)) return -10;
    FieldView v;
    int rc = field_view(c, field, level, &v);
    if (rc) return rc;
    if (first_row < 0 || nrows < 1 || first_row + nrows > (int64_t)v.rows || x0 < 0 || ncols < 1 || x0 + ncols > c->nx) {
        set_error("fdtd_read_window: the window must lie inside the slab");
        return -3;
    }
    return copy_rows(c, dst, v.ptr + (size_t)first_row * c->pitch + x0, (size_t)nrows, (size_t)ncols, true);
}

int fdtd_write_field(fdtd_ctx *c, int field, int level, const float *src, size_t count)
{
    if (use_device(c)) return -10;
    FieldView v;
    int rc = field_view(c, field, level, &v);
    if (rc) return rc;
    if (count != v.rows * c->nx) {
        set_error("fdtd_write_field: count must be the number of cells of the slab");
        return -3;
    }
    c->halo_stale = true;
    return copy_rows(c, const_cast<float *>(src), v.ptr, v.rows, (size_t)c->nx, false);
}

int fdtd_snapshot_wait(fdtd_ctx *c)
{
    if (!c->snap[0].pending && !c->snap[1].pending) return 0;
    if (use_device(c)) return -10;
    HIPCHK(hipStreamSynchronize(c->side));
    // oldest first: snap_next is the set the next snapshot will take
    for (int i = 0; i < 2; ++i) {
        fdtd_ctx::Snap &sn = c->snap[(c->snap_next + i) & 1];
        if (!sn.pending) continue;
        if (sn.dst) std::memcpy(sn.dst, sn.host, sn.count * sizeof(float));
        sn.pending = false;
    }
    return 0;
}

const float *fdtd_snapshot_data(fdtd_ctx *c)
{
    if (c->snap_last < 0 || c->snap[c->snap_last].pending) return nullptr;
    return c->snap[c->snap_last].host;
}

int fdtd_snapshot_async(fdtd_ctx *c, int field, float *dst, size_t count)
{
    if (use_device(c)) return -10;
    FieldView v;
    int rc = field_view(c, field, 0, &v);
    if (rc) return rc;
    if (count != v.rows * c->nx) {
        set_error("fdtd_snapshot_async: count must be the number of cells of the slab");
        return -3;
    }
    fdtd_ctx::Snap &sn = c->snap[c->snap_next];
    if (sn.pending) {
        // both staging sets are on their way: wait for the older one (this one)
        if (c->snap_guard_at >= 0 && c->snap_guard == c->snap_next) c->snap_guard_at = -1;
        HIPCHK(hipEventSynchronize(sn.done));
        if (sn.dst) std::memcpy(sn.dst, sn.host, sn.count * sizeof(float));
        sn.pending = false;
    }
    if (sn.count != count) {
        HIPCHK(hipStreamSynchronize(c->side));
        if (sn.dev) (void)hipFree(sn.dev);
        if (sn.host) (void)hipHostFree(sn.host);
        sn.dev = nullptr;
        sn.host = nullptr;
        sn.count = 0;
        HIPCHK(hipMalloc((void **)&sn.dev, count * sizeof(float)));
        HIPCHK(hipHostMalloc((void **)&sn.host, count * sizeof(float), hipHostMallocDefault));
        sn.count = count;
    }
    if (!sn.copied) {
        HIPCHK(hipEventCreateWithFlags(&sn.copied, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&sn.done, hipEventDisableTiming));
    }
    // Side stream, behind the steps queued so far: a compact device copy, then the slow D2H.  The
    // one-pass kernels write the arrays they do not read, so the next sweep leaves the source as
    // it is and runs beside the device copy; the sweep after it waits for `copied`.  The split
    // kernels update H in place: there the very next sweep waits.
    HIPCHK(hipEventRecord(c->snap_ev, c->stream));
    HIPCHK(hipStreamWaitEvent(c->side, c->snap_ev, 0));
    HIPCHK(hipMemcpy2DAsync(sn.dev, (size_t)c->nx * 4, v.ptr, (size_t)c->pitch * 4, (size_t)c->nx * 4, v.rows,
                            hipMemcpyDeviceToDevice, c->side));
    HIPCHK(hipEventRecord(sn.copied, c->side));
    HIPCHK(hipMemcpyAsync(sn.host, sn.dev, count * sizeof(float), hipMemcpyDeviceToHost, c->side));
    HIPCHK(hipEventRecord(sn.done, c->side));
    if (c->snap_guard_at >= 0) {
        // an earlier guard is still open: the main stream takes it now
        HIPCHK(hipStreamWaitEvent(c->stream, c->snap[c->snap_guard].copied, 0));
    }
    c->snap_guard = c->snap_next;
    c->snap_guard_at = c->launches + (c->one_pass() ? 1 : 0);
    sn.dst = dst;
    sn.pending = true;
    c->snap_last = c->snap_next;
    c->snap_next ^= 1;
    return 0;
}

int fdtd_probe_fields(const fdtd_ctx *c) { return c->is2d() ? 4 : 9; }

int fdtd_set_probes(fdtd_ctx *c, int count, const int32_t *xyz, int capacity)
{
    if (count < 0 || count > FDTD_MAX_PROBES || capacity < 1) {
        set_error("fdtd_set_probes: 0..16 probes, capacity >= 1");
        return -3;
    }
    if (use_device(c)) return -10;
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->probe_ring) (void)hipFree(c->probe_ring);
    c->probe_ring = nullptr;
    c->probes.count = count;
    for (int i = 0; i < count; ++i) {
        c->probes.x[i] = xyz[i * 3];
        c->probes.y[i] = xyz[i * 3 + 1];
        c->probes.z[i] = c->is2d() ? 0 : xyz[i * 3 + 2];
    }
    c->probe_cap = capacity;
    c->probe_first = c->step;
    const size_t n = (size_t)capacity * FDTD_MAX_PROBES * fdtd_probe_fields(c);
    HIPCHK(hipMalloc((void **)&c->probe_ring, n * sizeof(float)));
    HIPCHK(hipMemsetAsync(c->probe_ring, 0, n * sizeof(float), c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int fdtd_read_probes(fdtd_ctx *c, int64_t first_step, int nsteps, float *dst, size_t count)
{
    if (use_device(c)) return -10;
    const int nf = fdtd_probe_fields(c), np = c->probes.count;
    if (!c->probe_ring || first_step < c->probe_first || first_step + nsteps > c->step ||
        count != (size_t)nsteps * np * nf) {
        set_error("fdtd_read_probes: rows not in the ring, or wrong count");
        return -3;
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    // the rows sit in a ring: one copy, two when the range wraps around its end
    const size_t row = (size_t)FDTD_MAX_PROBES * nf;
    std::vector<float> tmp((size_t)nsteps * row);
    const int first_slot = (int)(first_step % c->probe_cap);
    const int n1 = std::min(nsteps, c->probe_cap - first_slot);
    if (n1 > 0)
        HIPCHK(hipMemcpy(tmp.data(), c->probe_ring + (size_t)first_slot * row, (size_t)n1 * row * sizeof(float),
                         hipMemcpyDeviceToHost));
    if (nsteps > n1)
        HIPCHK(hipMemcpy(tmp.data() + (size_t)n1 * row, c->probe_ring, (size_t)(nsteps - n1) * row * sizeof(float),
                         hipMemcpyDeviceToHost));
    for (int s = 0; s < nsteps; ++s)
        std::memcpy(dst + (size_t)s * np * nf, tmp.data() + (size_t)s * row, (size_t)np * nf * sizeof(float));
    return 0;
}

size_t fdtd_halo_count(const fdtd_ctx *c)
{
    if (c->deep()) return c->deep_count();
    return c->is2d() ? (size_t)c->nx : (size_t)3 * c->nx * c->ny;
}

static int halo_copy(fdtd_ctx *c, int side, float *host, size_t count, bool get, float *staging)
{
    if (side < 0 || side > 1 || count != fdtd_halo_count(c)) {
        set_error("halo: bad side or count");
        return -3;
    }
    if (use_device(c)) return -10;
    if (c->deep()) {
        int rc;
        if (get) {
            float *buf = staging;
            if (!buf) {
                if ((rc = ensure_deep_buffers(c))) return rc;
                buf = c->halo_send[side];
            }
            if ((rc = deep_halo_kernel(c, 1 << side, false, buf, buf, c->stream))) return rc;
            HIPCHK(hipMemcpyAsync(host, buf, count * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        } else {
            if ((rc = ensure_deep_buffers(c))) return rc;
            HIPCHK(hipMemcpyAsync(c->halo_recv[side], host, count * sizeof(float), hipMemcpyHostToDevice, c->stream));
            if ((rc = deep_halo_kernel(c, 1 << side, true, c->halo_recv[side], c->halo_recv[side], c->stream)))
                return rc;
            // the neighbour's material codes travel with the rows: they decide the kernel variant
            const size_t n = (size_t)c->tb_steps * c->nx;
            HIPCHK(hipStreamSynchronize(c->stream));
            if ((rc = note_halo_codes(c, side, (const uint8_t *)(host + (size_t)6 * n)))) return rc;
        }
        HIPCHK(hipStreamSynchronize(c->stream));
        return 0;
    }
    const int c0 = c->is2d() ? 2 : 0;
    const size_t rows = c->is2d() ? 1 : (size_t)c->ny;
    for (int comp = c0; comp < 3; ++comp) {
        float *dev = halo_rows(c, comp, side, !get);
        float *h = host + (size_t)(comp - c0) * c->nx * rows;
        if (get)
            HIPCHK(hipMemcpy2DAsync(h, (size_t)c->nx * 4, dev, (size_t)c->pitch * 4, (size_t)c->nx * 4, rows,
                                    hipMemcpyDeviceToHost, c->stream));
        else
            HIPCHK(hipMemcpy2DAsync(dev, (size_t)c->pitch * 4, h, (size_t)c->nx * 4, (size_t)c->nx * 4, rows,
                                    hipMemcpyHostToDevice, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int fdtd_halo_get(fdtd_ctx *c, int side, float *dst, size_t count)
{
    return halo_copy(c, side, dst, count, true, nullptr);
}
int fdtd_halo_get_noalloc(fdtd_ctx *c, int side, float *dst, size_t count)
{
    // Only the packed block of the multi-step mode is staged; the one-row (2-D) and one-plane (3-D)
    // forms copy straight out of the field arrays, whatever their size.
    if (!c->deep()) return halo_copy(c, side, dst, count, true, nullptr);
    if (count > kScratchFloats) {
        set_error("fdtd_halo_get_noalloc: the block exceeds the 4 MiB (1048576 floats) of built-in staging memory; "
                  "use fdtd_halo_get");
        return -3;
    }
    if (use_device(c)) return -10;
    float *stage = scratch_of(c->device);
    if (!stage) {
        set_error("fdtd_halo_get_noalloc: no staging memory on this device (hipGetSymbolAddress failed at fdtd_create)");
        return -10;
    }
    if (count > 0) {
        // the last float holds the tail of the code bytes and may be written in part only; the
        // array keeps what the previous call (of any context on this GPU) left there
        HIPCHK(hipMemsetAsync(stage + count - 1, 0, sizeof(float), c->stream));
    }
    return halo_copy(c, side, dst, count, true, stage);
}

int fdtd_halo_put(fdtd_ctx *c, int side, const float *src, size_t count)
{
    return halo_copy(c, side, const_cast<float *>(src), count, false, nullptr);
}

// ---- checkpoint ----------------------------------------------------------
// Raw binary image of the time-dependent state: a header, then every field
// array as it is stored (halo rows and padding included) in a fixed order.

} // extern "C"

namespace {

struct CheckpointHeader {
    char magic[8]; // "FDTDCKP2"
    int32_t mode, kernel, ecoef, nx, ny, nz, s0, sn, pitch, cur, hgen, narrays, pair, halo;
    int64_t step, elems;
};

// Every allocation that holds time-dependent state, in the order it was made (c->elems floats each).
std::vector<float *> state_arrays(fdtd_ctx *c)
{
    std::vector<float *> v;
    for (const auto &b : c->state_bufs) v.push_back(b.first);
    return v;
}

CheckpointHeader header_of(fdtd_ctx *c, int narrays)
{
    CheckpointHeader h;
    std::memset(&h, 0, sizeof h);
    std::memcpy(h.magic, "FDTDCKP2", 8);
    h.mode = c->mode;
    h.kernel = c->kernel;
    h.ecoef = c->ecoef;
    h.nx = c->nx;
    h.ny = c->ny;
    h.nz = c->nz;
    h.s0 = c->s0;
    h.sn = c->sn;
    h.pitch = c->pitch;
    h.cur = c->cur;
    h.hgen = c->hgen;
    h.pair = c->pair;
    h.halo = c->halo;
    h.narrays = narrays;
    h.step = c->step;
    h.elems = (int64_t)c->elems;
    return h;
}

constexpr size_t kChunk = (size_t)16 << 20; // floats per staging chunk (64 MB)

} // namespace

extern "C" {

int fdtd_checkpoint_save(fdtd_ctx *c, const char *path)
{
    if (use_device(c)) return -10;
    HIPCHK(hipStreamSynchronize(c->stream));
    const std::vector<float *> arrays = state_arrays(c);
    FILE *f = std::fopen(path, "wb");
    if (!f) {
        set_error(std::string("Couldn't open file ") + path);
        return -1;
    }
    const CheckpointHeader h = header_of(c, (int)arrays.size());
    bool ok = std::fwrite(&h, sizeof h, 1, f) == 1;
    std::vector<float> stage(std::min(kChunk, c->elems));
    for (size_t a = 0; a < arrays.size() && ok; ++a)
        for (size_t o = 0; o < c->elems && ok; o += kChunk) {
            const size_t n = std::min(kChunk, c->elems - o);
            if (hipMemcpy(stage.data(), arrays[a] + o, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) {
                std::fclose(f);
                set_error("fdtd_checkpoint_save: device read failed");
                return -10;
            }
            ok = std::fwrite(stage.data(), sizeof(float), n, f) == n;
        }
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) {
        set_error(std::string("fdtd_checkpoint_save: write to ") + path + " failed");
        return -2;
    }
    return 0;
}

int fdtd_checkpoint_load(fdtd_ctx *c, const char *path)
{
    if (use_device(c)) return -10;
    HIPCHK(hipStreamSynchronize(c->stream));
    const std::vector<float *> arrays = state_arrays(c);
    FILE *f = std::fopen(path, "rb");
    if (!f) {
        set_error(std::string("Couldn't open file ") + path);
        return -1;
    }
    CheckpointHeader h, want = header_of(c, (int)arrays.size());
    if (std::fread(&h, sizeof h, 1, f) != 1 || std::memcmp(h.magic, want.magic, 8) != 0) {
        std::fclose(f);
        set_error(std::string(path) + " is not a checkpoint file");
        return -2;
    }
    if (h.mode != want.mode || h.kernel != want.kernel || h.ecoef != want.ecoef || h.nx != want.nx ||
        h.ny != want.ny || h.nz != want.nz || h.s0 != want.s0 || h.sn != want.sn || h.pitch != want.pitch ||
        h.narrays != want.narrays || h.elems != want.elems || h.halo != want.halo) {
        std::fclose(f);
        set_error("fdtd_checkpoint_load: the checkpoint was written by a context of another shape "
                  "(grid, slab, mode, kernel or ecoef differ)");
        return -3;
    }
    std::vector<float> stage(std::min(kChunk, c->elems));
    for (size_t a = 0; a < arrays.size(); ++a)
        for (size_t o = 0; o < c->elems; o += kChunk) {
            const size_t n = std::min(kChunk, c->elems - o);
            if (std::fread(stage.data(), sizeof(float), n, f) != n) {
                std::fclose(f);
                set_error(std::string(path) + " is truncated");
                return -2;
            }
            if (hipMemcpy(arrays[a] + o, stage.data(), n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
                std::fclose(f);
                set_error("fdtd_checkpoint_load: device write failed");
                return -10;
            }
        }
    std::fclose(f);
    c->cur = h.cur;
    c->pair = h.pair;
    c->hgen = h.hgen;
    c->step = h.step;
    c->probe_first = c->step;
    c->halo_stale = true;
    return 0;
}

int fdtd_comm_unique_id(char id[128]) { return comm_unique_id(id); }

int fdtd_comm_init(fdtd_ctx *c, const char id[128], int rank, int world)
{
    if (world < 1 || rank < 0 || rank >= world) {
        set_error("fdtd_comm_init: bad rank/world");
        return -3;
    }
    if (use_device(c)) return -10;
    if (c->comm) {
        comm_destroy(c->comm);
        c->comm = nullptr;
    }
    c->rank = rank;
    c->world = world;
    c->halo_stale = true;
    return comm_create(id, rank, world, &c->comm);
}

int fdtd_comm_local_group(int world, void **group)
{
    if (!group || world < 1) {
        set_error("fdtd_comm_local_group: bad argument");
        return -3;
    }
    *group = local_group_create(world);
    return 0;
}

void fdtd_comm_local_group_free(void *group) { local_group_destroy((LocalGroup *)group); }

int fdtd_comm_init_local(fdtd_ctx *c, void *group, int rank, int world)
{
    if (!group || world < 1 || rank < 0 || rank >= world) {
        set_error("fdtd_comm_init_local: bad group, rank or world");
        return -3;
    }
    if (use_device(c)) return -10;
    if (c->comm) {
        comm_destroy(c->comm);
        c->comm = nullptr;
    }
    c->rank = rank;
    c->world = world;
    c->halo_stale = true;
    return comm_create_local((LocalGroup *)group, rank, &c->comm);
}

int fdtd_comm_selftest(fdtd_ctx *c)
{
    if (!c->comm) {
        set_error("fdtd_comm_selftest: call fdtd_comm_init first");
        return -3;
    }
    if (use_device(c)) return -10;
    if (c->deep()) {
        // the packed boundary rows of side 0 -> the receive buffer of side 1, through ncclSend/ncclRecv
        const size_t n = c->deep_count();
        int rc = ensure_deep_buffers(c);
        if (!rc) rc = deep_halo_kernel(c, 3, false, c->halo_send[0], c->halo_send[1], c->stream);
        if (!rc) rc = comm_group_start(c->comm);
        if (!rc) rc = comm_send(c->comm, c->halo_send[0], n, c->rank, c->stream);
        if (!rc) rc = comm_recv(c->comm, c->halo_recv[1], n, c->rank, c->stream);
        const int rc2 = comm_group_end(c->comm);
        if (rc || rc2) return rc ? rc : rc2;
        std::vector<float> a(n), b(n);
        HIPCHK(hipMemcpyAsync(a.data(), c->halo_send[0], n * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(b.data(), c->halo_recv[1], n * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (std::memcmp(a.data(), b.data(), n * 4) != 0) {
            set_error("fdtd_comm_selftest: the received halo block differs from the block sent");
            return -22;
        }
        return 0;
    }
    // first owned row of the newest E level -> upper halo row, through ncclSend/ncclRecv to this rank
    const int comp = c->interleave3 ? 0 : 2; // interleaved components: the run that starts at ex holds all three
    const size_t count = c->is2d() ? (size_t)c->nx : (size_t)c->ny * c->pitch;
    int rc = comm_group_start(c->comm);
    if (!rc) rc = comm_send(c->comm, halo_rows(c, comp, 0, false), count, c->rank, c->stream);
    if (!rc) rc = comm_recv(c->comm, halo_rows(c, comp, 1, true), count, c->rank, c->stream);
    const int rc2 = comm_group_end(c->comm);
    if (rc || rc2) return rc ? rc : rc2;
    std::vector<float> a(count), b(count);
    HIPCHK(hipMemcpyAsync(a.data(), halo_rows(c, comp, 0, false), count * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(b.data(), halo_rows(c, comp, 1, true), count * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (std::memcmp(a.data(), b.data(), count * 4) != 0) {
        set_error("fdtd_comm_selftest: the received row differs from the row sent");
        return -22;
    }
    return 0;
}

} // extern "C"
