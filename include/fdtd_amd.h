/*
 * fdtd_amd.h -- C ABI of the MI355X (gfx950) FDTD field-update library.
 *
 * The reference (UnalignedByte/fdtd) has no plugin or FFI layer: its update
 * path sits behind `main` (fdtd.cu:46-544) and six symbols in
 * fdtd_calculations.h:18-42.  Each entry point below names the reference code
 * it replaces.  Conventions: plain pointers and sizes only; `int` return,
 * 0 = success, < 0 = error with the message available from fdtd_last_error();
 * the library never calls exit(); the caller owns host buffers, the library
 * owns every device buffer; one context must not be used from two threads at
 * once.  All field data is fp32, cell (x, y, z) of a full field lives at
 * z*ny*nx + y*nx + x (utils.h:15).
 */
#ifndef FDTD_AMD_H
#define FDTD_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FDTD_MAX_SOURCES 16 /* fdtd_calculations.h:7 */
#define FDTD_MAX_JZ 512     /* fdtd_calculations.h:8 */
#define FDTD_MAX_CODES 256  /* material codes; the reference table (fdtd.cu:878) has 94 and overflows */
#define FDTD_MAX_PROBES 16

/* Mirror of FdtdParams (fdtd.h:7-34) with fixed-size members.  jz holds the
 * FDTD_MAX_JZ samples that reach the kernels (fdtd_calculations.cu:27-28);
 * the sample index of step n is n - n % 3 (fdtd.cu:172,310,448) and samples
 * past the table are 0. */
typedef struct fdtd_params {
    int32_t nx, ny, nz;
    int32_t iterations;                 /* rounded up to a multiple of 3, fdtd.cu:563 */
    char input_path[1024];              /* env_set_dir: directory of the v1_%05d.pgm slices */
    char output_path[1024];             /* output_dir */
    int32_t elements_per_wave;
    float wave_frequency;
    float pulse_width;
    float pulse_modulation_frequency;
    int32_t sources_count;
    int32_t sources[FDTD_MAX_SOURCES * 3]; /* 0-based x, y, z (fdtd.cu:607-609) */
    float default_sigma, default_eps_s, default_eps_i, default_tau_d;
    int32_t jz_count;
    float jz[FDTD_MAX_JZ];
    float pi, c, timeskip, lambda, dt, dx, dy, dz, mu0, eps0;
} fdtd_params;

enum { FDTD_MODE_AUTO = 0, FDTD_MODE_3D = 1, FDTD_MODE_2D_TMZ = 2 };
/* d = 2b - sigma*dt^2 as written in fdtd_calculations.cu:144 (fields grow ~2x
 * per step in vacuum), or d = a + b - sigma*dt^2 (bounded). */
enum { FDTD_ECOEF_REFERENCE = 0, FDTD_ECOEF_CONSISTENT = 1 };
/* SPLIT: an H kernel and a fused D+source+E kernel (H updated in place).
 * FUSED: one sweep over memory per step (H double-buffered).
 * MULTISTEP (2-D TMz only): one sweep over memory per `steps_per_pass` steps;
 * the same arithmetic per cell and step, 16 more bytes of HBM per cell (E and D
 * keep four arrays instead of two).  fdtd_step(n) runs n / steps_per_pass
 * passes and the remainder as one shorter pass (a single step: FUSED).
 * TILE (2-D TMz only): storage, schedule and halo protocol of MULTISTEP, but a
 * workgroup advances a tile of 64 columns x 64 (32) rows `steps_per_pass`
 * steps out of registers and LDS and writes the Mur faces and the probe rows
 * itself: ONE launch per pass, for grids that are bound by launch and memory
 * latency instead of bandwidth.  steps_per_pass 1 .. 8.
 * AUTO: 2-D grids of 7.5 M cells or more MULTISTEP, smaller ones TILE (extents of
 * the WHOLE grid, so every slab of it decides alike), 3-D FUSED; a whole grid
 * whose MULTISTEP arrays do not fit into device memory falls back to FUSED, a
 * slab reports the error. */
enum { FDTD_KERNEL_AUTO = 0, FDTD_KERNEL_SPLIT = 1, FDTD_KERNEL_FUSED = 2, FDTD_KERNEL_MULTISTEP = 3, FDTD_KERNEL_TILE = 4 };

typedef struct fdtd_options {
    int32_t mode;        /* FDTD_MODE_*; AUTO = 2-D TMz when nz == 1 */
    int32_t ecoef;       /* FDTD_ECOEF_* */
    int32_t device;      /* HIP device ordinal */
    int32_t kernel;      /* FDTD_KERNEL_* */
    int32_t slab_begin;  /* first global row of the slab axis held here (y in 2-D, z in 3-D) */
    int32_t slab_count;  /* rows held; 0 = the whole grid */
    int32_t chunk_rows;  /* rows marched by one workgroup of the fused sweep; 0 = default */
    int32_t steps_per_pass; /* FDTD_KERNEL_MULTISTEP: 2 .. 8; 0 = default (8, 7 below 12 M cells; a whole grid drops to
                             * 7 once it holds material codes above 83, a slab or an explicit value never changes) */
    int32_t reserved[8];
} fdtd_options;

/* Field ids (read/write/snapshot/probe). */
enum {
    FDTD_HX = 0, FDTD_HY, FDTD_HZ,
    FDTD_DX, FDTD_DY, FDTD_DZ,
    FDTD_EX, FDTD_EY, FDTD_EZ,
    FDTD_NFIELDS
};
/* Mur coefficient planes (fdtd.h:52-53). */
enum { FDTD_RP_X0 = 0, FDTD_RP_Y0, FDTD_RP_Z0, FDTD_RP_XEND, FDTD_RP_YEND, FDTD_RP_ZEND };

typedef struct fdtd_ctx fdtd_ctx;

const char *fdtd_last_error(void);
void fdtd_options_default(fdtd_options *opt);

/* ---- input surface ---------------------------------------------------- */

/* initParamsWithPath (fdtd.cu:547-637) followed by setupSources
 * (fdtd.cu:1027-1082): positional parse of a data/input_params file, derived
 * constants, jz table. */
int fdtd_params_load(const char *path, fdtd_params *out);
/* fdtd.cu:625-634 on a struct filled in by the caller. */
int fdtd_params_derive(fdtd_params *p);
/* setupSources (fdtd.cu:1027-1082) on a struct whose constants are derived. */
int fdtd_setup_sources(fdtd_params *p);

/* Host-only pieces of the input surface, usable without a GPU. */

/* First half of loadMaterials (fdtd.cu:878-904): parse a mat_specs file into
 * table4[code*4 + {0,1,2,3}] = sigma, eps_s, eps_inf, tau_d (256 rows) and
 * present[code] = 1 for the codes the file defines. */
int fdtd_read_material_specs(const char *path, float *table4, uint8_t *present);
/* Second half of loadMaterials (fdtd.cu:909-936) for one slice file: the codes
 * of rows [y0, y0 + rows) of an nx x ny grid, x fastest, into dst. */
int fdtd_read_pgm_codes(const char *path, int nx, int ny, int upsample, int y0, int rows, uint8_t *dst);
/* The E-update coefficients of one material (fdtd_calculations.cu:141-156):
 * acd[3] = a, c, d (fp32) and k[3] = 2(dt+tau), 2dt+4tau, 2tau (fp64). */
int fdtd_material_coefficients(const fdtd_params *p, int ecoef, float sigma, float eps_s, float eps_inf,
                               float tau_d, float acd[3], double k[3]);
/* One value of an rp plane (fdtd.cu:959-963 for the x faces, :983-987 for the
 * y and z faces) for a material. */
float fdtd_rp_value(const fdtd_params *p, int x_face, float sigma, float eps_s, float eps_inf, float tau_d);

/* ---- context ---------------------------------------------------------- */

/* initDeviceFieldWithParams + copyDataToDevice + setupMurBoundary +
 * copySymbolsToDevice (fdtd.cu:764-822, 1085-1133, 942-1024;
 * fdtd_calculations.cu:7-29): allocates the padded SoA fields on the device,
 * zeroes them, fills every cell with the default material and derives the rp
 * planes from the defaults (the reference runs setupMurBoundary before
 * loadMaterials, fdtd.cu:59 vs :65). */
int fdtd_create(const fdtd_params *p, const fdtd_options *opt, fdtd_ctx **out);
void fdtd_destroy(fdtd_ctx *ctx);

/* loadMaterials (fdtd.cu:875-939): mat_specs file + one P2 PGM per z slice,
 * `<pgm_dir>/v1_%05d.pgm`, nx*ny tokens read in sequence.  first_slice is the
 * 1-based index of the slice for z = 0 (the reference always uses 1);
 * upsample >= 1 repeats each PGM pixel upsample x upsample times in x and y
 * (nearest neighbour; 1 = the reference behaviour); with upsample > 1 the
 * slice images go to the device as read and a kernel expands them into the
 * per-cell code map. */
int fdtd_load_materials(fdtd_ctx *ctx, const char *specs_path, const char *pgm_dir,
                        int first_slice, int upsample);
/* The two halves of loadMaterials on caller data: one table row per material
 * code (sigma, eps_s, eps_inf, tau_d; fdtd.cu:897-900) and the per-cell code
 * map (fdtd.cu:922-931; `count` = cells of the slab held, x fastest). */
int fdtd_set_material(fdtd_ctx *ctx, int code, float sigma, float eps_s, float eps_inf, float tau_d);
int fdtd_set_codes(fdtd_ctx *ctx, const uint8_t *codes, size_t count);
/* Replace one rp plane (layouts of fdtd.cu:952-1001, global extents: `count`
 * is the size of the whole plane whatever the context holds).  A context keeps
 * the entries its kernels read -- a slab those of its own rows or planes and of
 * the neighbours' rows a multi-step pass recomputes, a 2-D context none of the
 * z faces -- so the memory a plane takes follows the slab, not the grid.
 * fdtd_get_rp() returns the held entries and, everywhere else, the value
 * derived from the defaults. */
int fdtd_set_rp(fdtd_ctx *ctx, int face, const float *data, size_t count);
int fdtd_get_rp(fdtd_ctx *ctx, int face, float *data, size_t count);
/* setupMurBoundary (fdtd.cu:942-1024) evaluated on the materials now loaded
 * (opt-in: the reference only ever evaluates it on the defaults).  A slab
 * fills in the entries of the cells it holds, which are the only ones its Mur
 * updates read; the other entries of its planes keep the default-derived
 * values (fdtd_get_rp returns them as such).  A slab of the multi-step mode
 * also recomputes the neighbours' rows in its halos, Mur included: their x-face
 * entries are taken from the material codes that arrive with the next halo
 * exchange, which this call makes due (host-staged exchange: do it before the
 * next fdtd_step). */
int fdtd_setup_mur_from_materials(fdtd_ctx *ctx);

/* ---- time stepping ---------------------------------------------------- */

/* The hot loop (fdtd.cu:115-530): nsteps x { updateHField, updateDField,
 * updateSources, updateEField, updateMurBoundary } (fdtd_calculations.cu:32-428).
 * Asynchronous: returns once the work is queued on the context's stream. */
int fdtd_step(fdtd_ctx *ctx, int nsteps);
int fdtd_sync(fdtd_ctx *ctx);
int64_t fdtd_steps_done(const fdtd_ctx *ctx);
/* Time steps one launch of the main sweep advances (1 unless MULTISTEP). */
int fdtd_steps_per_pass(const fdtd_ctx *ctx);

/* Per-step device timing of the field sweep with HIP events on the stream the
 * kernels run on.  After fdtd_profile_begin(ctx, capacity) each step records
 * an event pair around its main sweep; fdtd_profile_read() synchronises and
 * returns the milliseconds of up to `capacity` steps. */
int fdtd_profile_begin(fdtd_ctx *ctx, int capacity);
int fdtd_profile_read(fdtd_ctx *ctx, float *sweep_ms, int capacity, int *count);
/* The same, and steps[i] = the time steps launch i advanced (1 for the one-step
 * kernels; a pass of the multi-step sweep advances up to steps_per_pass, the
 * remainder pass of an fdtd_step() call fewer). */
int fdtd_profile_read_launches(fdtd_ctx *ctx, float *sweep_ms, int32_t *steps, int capacity, int *count);
/* Wave timeline of ONE launch of the multi-step 2-D sweep (diagnostics; no
 * counterpart in the reference): fdtd_profile_waves_begin() arms it for the
 * next pass with room for `capacity` waves, fdtd_profile_waves_read()
 * synchronises and returns four 64-bit words per launch position: start and
 * end of the wave on the GPU's constant 100 MHz clock, strip | band << 32,
 * and flags (bit 0: x guards folded away, bit 1: y guards folded away, bits
 * 8-11: XCC id, bits 32-63: the HW_ID register).  Positions of the padded last
 * workgroup stay zero.  Other kernels record nothing (count = 0). */
int fdtd_profile_waves_begin(fdtd_ctx *ctx, int64_t capacity);
int fdtd_profile_waves_read(fdtd_ctx *ctx, uint64_t *dst, int64_t capacity, int64_t *count);
/* Shape of one launch of the main sweep over the slab: the waves it starts and
 * the row iterations they execute in all (a wave of the multi-step sweep
 * marches its rows plus 2 * steps_per_pass - 1 rows around them, the first
 * 2 * (steps_per_pass - 1) of which -- away from the y faces -- run half of the
 * stages on average and count as steps_per_pass - 1; the one-step kernels
 * handle one row per wave).  Input of issue-rate estimates. */
int fdtd_sweep_work(const fdtd_ctx *ctx, int64_t *waves, int64_t *row_iterations);
/* Name of the kernel that does the main sweep (as a profiler reports it). */
const char *fdtd_sweep_kernel_name(const fdtd_ctx *ctx);

/* ---- results ---------------------------------------------------------- */

/* Synchronous copy of one field of the slab held (newest time level) to or
 * from host memory, unpadded, `count` = nx * rows * (ny or 1) floats.  This is
 * what the cudaMemcpyAsync + copy threads of fdtd.cu:134-136,176-178,221-223
 * deliver.  level 0 = newest, 1 = previous time level (E and D only). */
int fdtd_read_field(fdtd_ctx *ctx, int field, int level, float *dst, size_t count);
int fdtd_write_field(fdtd_ctx *ctx, int field, int level, const float *src, size_t count);
/* A rectangle of the same data: rows first_row .. first_row + nrows - 1 of the
 * slab (2-D: local y; 3-D: local z * ny + y, so one z slice -- the unit the
 * writers of fdtd.cu:1288-1494 put into one file -- is ny consecutive rows),
 * columns x0 .. x0 + ncols - 1, packed into dst[nrows][ncols]. */
int fdtd_read_window(fdtd_ctx *ctx, int field, int level, int64_t first_row, int64_t nrows, int x0, int ncols,
                     float *dst);

/* Snapshot of one field on a side stream: a device-to-device copy ordered
 * after the steps queued so far, then an async D2H into pinned staging; the
 * steps queued afterwards do not wait for the D2H.  fdtd_snapshot_wait()
 * blocks until the data is in dst. */
int fdtd_snapshot_async(fdtd_ctx *ctx, int field, float *dst, size_t count);
int fdtd_snapshot_wait(fdtd_ctx *ctx);
/* With dst = NULL the snapshot stays in the library's pinned staging buffer:
 * after fdtd_snapshot_wait() this returns it (count floats, valid until the
 * next fdtd_snapshot_async) and saves the copy into caller memory. */
const float *fdtd_snapshot_data(fdtd_ctx *ctx);

/* Probe time series: up to FDTD_MAX_PROBES cells (global 0-based x, y, z);
 * after every step the values of hx, hy, dz, ez (2-D) or all nine fields (3-D)
 * at the probes held by this slab are appended to a device ring of `capacity`
 * steps.  fdtd_read_probes() copies rows [first_step, first_step + nsteps) as
 * [step][probe][field] floats; probes outside the slab read as 0. */
int fdtd_set_probes(fdtd_ctx *ctx, int count, const int32_t *xyz, int capacity);
int fdtd_probe_fields(const fdtd_ctx *ctx);
int fdtd_read_probes(fdtd_ctx *ctx, int64_t first_step, int nsteps, float *dst, size_t count);

/* Checkpoint / restart (the reference has none: it dumps every step as text
 * and never reads anything back).  fdtd_checkpoint_save() writes the
 * time-dependent state of the slab -- all field arrays, time level indices and
 * the step counter -- as one raw binary file; fdtd_checkpoint_load() restores
 * it into a context created with the same parameters and options.  Materials,
 * rp planes and probes are inputs, not state: set them up as before. */
int fdtd_checkpoint_save(fdtd_ctx *ctx, const char *path);
int fdtd_checkpoint_load(fdtd_ctx *ctx, const char *path);

/* ---- slabs across GPUs ------------------------------------------------ */

/* Halo rows of the slab axis.  side 0 = towards lower indices, 1 = towards
 * higher.  fdtd_halo_get() copies the rows a neighbour needs (the first or
 * last owned rows of the newest E level) to the host; fdtd_halo_put() stores
 * rows received from that neighbour into the halo.  `count` must equal
 * fdtd_halo_count().  These are the host-staged form of the exchange. */
size_t fdtd_halo_count(const fdtd_ctx *ctx);
int fdtd_halo_get(fdtd_ctx *ctx, int side, float *dst, size_t count);
int fdtd_halo_put(fdtd_ctx *ctx, int side, const float *src, size_t count);
/* No counterpart in the reference.  For callers that cannot allocate device
 * memory (tight serving loops, harnesses that forbid hipMalloc):
 * fdtd_halo_get() with the packed block of the multi-step mode staged in a
 * 4 MiB array that is part of the library's code object instead of buffers
 * allocated on first use.  Same arguments and result as fdtd_halo_get(); the
 * packed block must not exceed 1048576 floats (the one-row and one-plane forms
 * of the other kernels need no staging and have no limit).  Like
 * fdtd_halo_get() it synchronises the stream and copies to pageable host
 * memory, so it cannot be captured into a HIP graph.  The array is one per GPU
 * and shared by every context on it: two calls that overlap on one GPU race. */
int fdtd_halo_get_noalloc(fdtd_ctx *ctx, int side, float *dst, size_t count);

/* RCCL form of the exchange: rank 0 creates an id (128 bytes), every rank
 * passes it to fdtd_comm_init(); afterwards fdtd_step() exchanges halos with
 * ranks rank-1 and rank+1 after every step (every pass of the multi-step
 * mode), ncclSend/ncclRecv in one group.  The rows next to a neighbour are
 * swept first on a second stream and the exchange follows them there while the
 * main stream sweeps the interior.  The slab of rank r must be the r-th slab.
 * Calls that change the state behind the stepper's back (fdtd_set_codes,
 * fdtd_load_materials, fdtd_write_field, fdtd_checkpoint_load, fdtd_comm_init)
 * make the next fdtd_step() of the multi-step mode start with an extra
 * exchange: every rank must make them between the same two fdtd_step() calls. */
int fdtd_comm_unique_id(char id[128]);
int fdtd_comm_init(fdtd_ctx *ctx, const char id[128], int rank, int world);
/* The same exchange between contexts of ONE process (no counterpart in the
 * reference, no RCCL): each rank is a context driven by its own host thread,
 * on one GPU or on several; messages are device-to-device copies after a
 * rendezvous of the two threads.  fdtd_comm_local_group() creates the shared rendezvous object for
 * `world` ranks, every rank passes it to fdtd_comm_init_local(); free it after
 * the contexts are destroyed.  An exchange that fails on any rank (time limit,
 * ranks that disagree on the message size, a HIP error) leaves the group
 * unusable: every later exchange through it reports an error. */
int fdtd_comm_local_group(int world, void **group);
void fdtd_comm_local_group_free(void *group);
int fdtd_comm_init_local(fdtd_ctx *ctx, void *group, int rank, int world);
/* Diagnostic: sends the first owned ez row (3-D: ez plane) to this very rank
 * through ncclSend/ncclRecv into the upper halo and checks that it arrived.
 * Usable with a one-rank communicator; overwrites the upper halo. */
int fdtd_comm_selftest(fdtd_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
