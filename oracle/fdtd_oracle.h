/*
 * fdtd_oracle.h -- CPU restatement of the UnalignedByte/fdtd field-update path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the checker the HIP path is compared with;
 * nothing in the shipped library may include, link or call it.  Only tests/,
 * __graft_entry__.smoke() and the cpu_baseline leg of bench.py use it.
 *
 * Pinning status.  The reference has no tests, golden vectors or fixtures and
 * its executable cannot be built here (nvcc and the CUDA runtime are absent).
 * Two things pin this file:
 *  - the reference's OWN kernel file, fdtd_calculations.cu, translated by the
 *    image's hipify-perl (cuda* runtime names -> hip*, nothing else) and built for
 *    gfx950 with -ffp-contract=off by `make ref` into _ref/libfdtd_ref.so
 *    (ref_driver.hip launches the five kernels in main's order and rotation):
 *    tests/test_reference_kernels_gpu.py compares the 3-D update -- H, D,
 *    sources, Debye E update, all twelve Mur cases, every phase of the buffer
 *    rotation -- with it on the GPU, every cell of every buffer bit for bit;
 *  - the known-answer values recorded when the reference was surveyed (derived
 *    constants, jz anchors, rp planes, an 11-row probe series of the shipped
 *    100^3 vacuum case): tests/golden/survey_known_answers.json and
 *    tests/test_oracle_golden.py (CPU only).  They are the only pins of the
 *    input surface (parameter parse, source waveform, rp formula).
 * The 2-D TMz reduction (the reference has no 2-D mode: with nz = 1 none of its
 * guards passes) is pinned step by step: one step of it is what the reference's
 * kernels do to the middle plane of a three-plane grid whose state does not
 * depend on z (same test file, last test).
 * UNPINNED: the CONSISTENT coefficient variant (one expression, d = a + b -
 * sigma dt^2), which is defined by this restatement.
 *
 * Every function cites the reference file:line it follows.  Arithmetic follows
 * the reference's C promotions (fp32 fields, fp64 in-flight in the E update)
 * and must be compiled with -ffp-contract=off.
 */
#ifndef FDTD_ORACLE_H
#define FDTD_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_SOURCES 16   /* fdtd_calculations.h:7  D_MAX_SOURCES */
#define ORC_MAX_JZ 512       /* fdtd_calculations.h:8  D_MAX_JZ      */
#define ORC_JZ_TABLE 16384   /* fdtd.cu:1076 entries produced by setupSources */
#define ORC_MAX_CODES 256    /* fdtd.cu:878 has 94 and overflows; see orc_load_materials */

enum { ORC_ECOEF_REFERENCE = 0, ORC_ECOEF_CONSISTENT = 1 };
enum { ORC_MODE_3D = 0, ORC_MODE_2D_TMZ = 1 };

/* Field ids for orc_field()/orc_set_field(). */
enum {
    ORC_HX = 0, ORC_HY, ORC_HZ,
    ORC_DX, ORC_DY, ORC_DZ,
    ORC_EX, ORC_EY, ORC_EZ,
    ORC_NFIELDS
};

/* Mirror of FdtdParams (fdtd.h:7-34) with fixed-size members. */
typedef struct orc_params {
    int32_t nx, ny, nz;
    int32_t iterations;            /* rounded up to a multiple of 3, fdtd.cu:563 */
    char input_path[1024];
    char output_path[1024];
    int32_t elements_per_wave;
    float wave_frequency;
    float pulse_width;
    float pulse_modulation_frequency;
    int32_t sources_count;
    int32_t sources[ORC_MAX_SOURCES * 3];   /* 0-based, fdtd.cu:607-609 */
    float default_sigma, default_eps_s, default_eps_i, default_tau_d;
    int32_t jz_count;
    float jz[ORC_JZ_TABLE];
    float pi, c, timeskip, lambda, dt, dx, dy, dz, mu0, eps0;
} orc_params;

typedef struct orc_state orc_state;

/* fdtd.cu:547-637: positional parse + derived constants.  0 on success. */
int orc_params_load(const char *path, orc_params *p);
/* fdtd.cu:625-634 alone (for parameters filled in by hand). */
void orc_params_derive(orc_params *p);
/* fdtd.cu:1027-1082.  0 on success, <0 where the reference would divide by
 * zero (fine == 0) or read an uninitialised istart. */
int orc_setup_sources(orc_params *p);

/* fdtd.cu:58-59,62,71: default materials everywhere, rp planes from the
 * defaults (setupMurBoundary runs before loadMaterials), all fields zero.
 * mode: ORC_MODE_3D or ORC_MODE_2D_TMZ (requires nz == 1).
 * Slab form (2-D only): the state holds global rows [gy0, gy0+nyl) of a grid
 * whose full height is p->ny, plus one halo row on either side.  gy0 = 0 and
 * nyl = p->ny is the whole grid. */
orc_state *orc_create(const orc_params *p, int mode, int ecoef);
orc_state *orc_create_slab2d(const orc_params *p, int ecoef, int gy0, int nyl);
void orc_destroy(orc_state *s);

/* fdtd.cu:875-939.  first_slice is the 1-based PGM index of z = 0 (the
 * reference always uses 1); 256-entry table instead of the reference's 94. */
int orc_load_materials(orc_state *s, const char *specs_path, const char *pgm_dir, int first_slice);
/* Per-cell material from a code map and a table of (sigma, eps_s, eps_i, tau_d)
 * rows -- the second half of loadMaterials (fdtd.cu:922-931) on caller data. */
int orc_set_materials(orc_state *s, const uint8_t *codes, const float *table4, int ncodes);
/* fdtd.cu:942-1024 evaluated on the CURRENT per-cell materials (the reference
 * only ever evaluates it on the defaults; this is the opt-in variant). */
void orc_setup_mur(orc_state *s);
/* Overwrite one rp plane: face 0..5 = x0,y0,z0,xEnd,yEnd,zEnd. */
int orc_set_rp(orc_state *s, int face, const float *data, size_t n);
float *orc_rp(orc_state *s, int face, size_t *n);

/* fdtd.cu:115-530: nsteps steps of H -> D -> sources -> E -> Mur with the
 * three-buffer rotation; threads > 1 splits the row loops with OpenMP. */
void orc_step(orc_state *s, int nsteps, int threads);
int orc_steps_done(const orc_state *s);

/* Pointer to a field at the newest time level (E, D) or the H arrays.
 * 3-D: nx*ny*nz floats, x fastest (utils.h:15).  2-D slab: (nyl+2) rows of nx,
 * row 0 being the lower halo; orc_field_rows() gives nyl+2. */
float *orc_field(orc_state *s, int field_id);
/* level 0 = newest (time n), 1 = n-1, 2 = n-2 buffer of the rotation. */
float *orc_field_level(orc_state *s, int field_id, int level);
size_t orc_field_count(const orc_state *s);

/* Per-cell material arrays in storage order: 0 sigma, 1 eps_s, 2 eps_i, 3 tau_d. */
float *orc_material(orc_state *s, int which);
/* The E-update coefficients a, b, c, d of one storage cell
 * (fdtd_calculations.cu:141-144). */
void orc_cell_coefs(const orc_state *s, size_t cell, float abcd[4]);

/* Timed run for the CPU baseline: returns seconds for nsteps steps. */
double orc_time_steps(orc_state *s, int nsteps, int threads);

#ifdef __cplusplus
}
#endif
#endif
