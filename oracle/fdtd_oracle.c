/*
 * fdtd_oracle.c -- CPU restatement of the UnalignedByte/fdtd field-update path.
 *
 * TEST INFRASTRUCTURE ONLY (see fdtd_oracle.h for who may use it and for the
 * pinning status).  Plain C, literal: per-cell material floats, three rotating
 * E/D buffers, one loop nest per reference kernel, the reference's guards and
 * quirks kept as they are.  Build with -ffp-contract=off.
 */
#include "fdtd_oracle.h"

#include <complex.h>
#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

struct orc_state {
    orc_params p;
    int mode, ecoef;
    int nx, ny, nz;      /* stored extents; 2-D slab: ny = nyl + 2 (halo rows) */
    int gy0, nyl, gny;   /* 2-D: first global row held, rows held, global rows */
    size_t n;
    float *h[3];
    float *d[3][3];      /* [buffer][component] */
    float *e[3][3];
    float *sigma, *eps_i, *eps_s, *tau_d;
    float *rp[6];        /* x0, y0, z0, xEnd, yEnd, zEnd */
    size_t rpn[6];
    int it;              /* steps done */
};

/* ------------------------------------------------------------------ text */

typedef struct {
    char *buf;
    size_t len, pos;
} cursor;

static int slurp(const char *path, cursor *c)
{
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    c->buf = (char *)malloc((size_t)sz + 1);
    c->len = fread(c->buf, 1, (size_t)sz, f);
    c->buf[c->len] = 0;
    c->pos = 0;
    fclose(f);
    return 0;
}

static void skip_ws(cursor *c)
{
    while (c->pos < c->len && isspace((unsigned char)c->buf[c->pos])) c->pos++;
}

/* One whitespace-delimited token ("%s"); NULL at end of input. */
static char *token(cursor *c, char *out, size_t cap)
{
    skip_ws(c);
    if (c->pos >= c->len) return NULL;
    size_t k = 0;
    while (c->pos < c->len && !isspace((unsigned char)c->buf[c->pos])) {
        if (k + 1 < cap) out[k++] = c->buf[c->pos];
        c->pos++;
    }
    out[k] = 0;
    return out;
}

/* "%d" / "%g": leading whitespace skipped, longest numeric prefix consumed. */
static int scan_int(cursor *c, int *v)
{
    skip_ws(c);
    char *end;
    long x = strtol(c->buf + c->pos, &end, 10);
    if (end == c->buf + c->pos) return -1;
    c->pos = (size_t)(end - c->buf);
    *v = (int)x;
    return 0;
}

static int scan_float(cursor *c, float *v)
{
    skip_ws(c);
    char *end;
    float x = strtof(c->buf + c->pos, &end);
    if (end == c->buf + c->pos) return -1;
    c->pos = (size_t)(end - c->buf);
    *v = x;
    return 0;
}

/* fgets(): consume through the next newline. */
static void skip_line(cursor *c)
{
    while (c->pos < c->len && c->buf[c->pos] != '\n') c->pos++;
    if (c->pos < c->len) c->pos++;
}

/* --------------------------------------------------------------- params */

/* fdtd.cu:625-634 */
void orc_params_derive(orc_params *p)
{
    p->pi = acos(-1.0);
    p->c = 3.0 * pow(10.0, 8.0);
    p->timeskip = 1.0;
    p->lambda = p->c / p->wave_frequency;
    p->dx = p->lambda / p->elements_per_wave;
    p->dy = p->dx;
    p->dz = p->dx;
    p->dt = 1.0 * p->timeskip /
            (p->c * sqrt(1.0 / pow(p->dx, 2.0) + 1.0 / pow(p->dy, 2.0) + 1.0 / pow(p->dz, 2.0)));
    p->mu0 = 4.0 * p->pi * pow(10.0, -7.0);
    p->eps0 = 1.0 / (p->mu0 * p->c * p->c);
}

/* fdtd.cu:547-637.  The reference parses by position: every key is read as a
 * token and dropped, a trailing "\n" in a scanf format eats all following
 * whitespace, and "unused" lines are skipped with fgets. */
int orc_params_load(const char *path, orc_params *p)
{
    cursor c;
    char key[1024];
    memset(p, 0, sizeof *p);
    if (slurp(path, &c) != 0) return -1;
    int rc = -2;

    if (!token(&c, key, sizeof key) || scan_int(&c, &p->nx) || scan_int(&c, &p->ny) ||
        scan_int(&c, &p->nz))
        goto done;                                             /* :560 */
    skip_ws(&c);
    if (!token(&c, key, sizeof key) || scan_int(&c, &p->iterations)) goto done; /* :562 */
    skip_ws(&c);
    p->iterations = ((p->iterations - 1) / 3 + 1) * 3;         /* :563 */
    skip_line(&c);                                             /* nf :565 */
    if (!token(&c, key, sizeof key) || !token(&c, p->input_path, sizeof p->input_path))
        goto done;                                             /* :567 */
    skip_ws(&c);
    skip_line(&c);                                             /* env_file_prefix :569 */
    if (!token(&c, key, sizeof key) || !token(&c, p->output_path, sizeof p->output_path))
        goto done;                                             /* :571 */
    skip_ws(&c);
    for (int i = 0; i < 9; i++) skip_line(&c);                 /* :573-589 */
    if (!token(&c, key, sizeof key) || scan_int(&c, &p->elements_per_wave)) goto done;   /* :591 */
    skip_ws(&c);
    if (!token(&c, key, sizeof key) || scan_float(&c, &p->wave_frequency)) goto done;    /* :593 */
    skip_ws(&c);
    if (!token(&c, key, sizeof key) || scan_float(&c, &p->pulse_width)) goto done;       /* :595 */
    skip_ws(&c);
    if (!token(&c, key, sizeof key) || scan_float(&c, &p->pulse_modulation_frequency))
        goto done;                                                                       /* :597 */
    skip_ws(&c);
    if (!token(&c, key, sizeof key) || scan_int(&c, &p->sources_count)) goto done;       /* :599 */
    skip_ws(&c);
    if (p->sources_count < 0 || p->sources_count > ORC_MAX_SOURCES) {
        rc = -3; /* only D_MAX_SOURCES reach the kernels, fdtd_calculations.cu:24 */
        goto done;
    }
    for (int i = 0; i < p->sources_count; i++) {               /* :602-610 */
        if (!token(&c, key, sizeof key)) goto done;
        for (int k = 0; k < 3; k++) {
            if (scan_int(&c, &p->sources[i * 3 + k])) goto done;
            p->sources[i * 3 + k] -= 1;
        }
        skip_ws(&c);
    }
    skip_line(&c);                                             /* pulse_type :612 */
    if (!token(&c, key, sizeof key) || scan_float(&c, &p->default_sigma)) goto done;     /* :614 */
    if (!token(&c, key, sizeof key) || scan_float(&c, &p->default_eps_s)) goto done;     /* :616 */
    if (!token(&c, key, sizeof key) || scan_float(&c, &p->default_eps_i)) goto done;     /* :618 */
    if (!token(&c, key, sizeof key) || scan_float(&c, &p->default_tau_d)) goto done;     /* :620 */
    orc_params_derive(p);
    rc = 0;
done:
    free(c.buf);
    return rc;
}

/* fdtd.cu:1027-1082 */
int orc_setup_sources(orc_params *p)
{
    int fine, temp, i2, istart = 0, have_start = 0;
    const int off = 1 << 16;
    float *jz = (float *)calloc((size_t)off, sizeof(float));
    float *tmp = (float *)calloc((size_t)off * 2, sizeof(float));
    float *tmp2 = (float *)calloc((size_t)off * 2, sizeof(float));
    int rc = 0;

    fine = (1 << 13) * p->pulse_width * p->wave_frequency * p->dt;          /* :1039 */
    if (fine <= 0) { rc = -1; goto done; }
    temp = 1.0 / (p->pulse_width * p->wave_frequency) / (p->dt / fine) / 2.0; /* :1040 */
    if (temp + 1 >= off) { rc = -2; goto done; }

    for (int i = -temp - 1; i <= temp + 1; i++) {                           /* :1043-1048 */
        float v1 = ((float)i / (((float)temp + 1.0) / 4.0));
        float v2 = exp(-pow(v1, 2.0));
        float v3 = cos(2.0 * acos(-1.0) * p->pulse_modulation_frequency * p->wave_frequency * i *
                       (p->dt / fine));
        tmp[i + off] = v2 * v3;
    }
    for (int i = -(1 << 12); i < (1 << 12); i++) {                          /* :1051-1056 */
        if ((fabsf(tmp[i + off]) >= 1e-9) && (i % fine == 0)) {
            istart = i;
            have_start = 1;
            break;
        }
    }
    if (!have_start) { rc = -3; goto done; }

    i2 = 0;
    for (int i = istart; i <= temp + 1; i += fine) {                        /* :1060-1064 */
        float val = tmp[i + off] * 1e-15 / p->dt / 3.0;
        jz[i2] = val;
        i2++;
    }
    for (int i = 2; i <= (1 << 14); i++) {                                  /* :1067-1073 */
        float val = (((jz[i] - jz[i - 1]) / p->dt) + ((jz[i - 1] - jz[i - 2]) / p->dt)) / 2.0 *
                    (p->dt * p->dz) / (p->dx * p->dy * p->dz);
        tmp2[i - 1 + off] = val;
    }
    for (int i = 0; i < ORC_JZ_TABLE; i++) p->jz[i] = tmp2[i + off + 1];    /* :1076-1078 */
    p->jz_count = ORC_JZ_TABLE;
done:
    free(jz);
    free(tmp);
    free(tmp2);
    return rc;
}

/* ---------------------------------------------------------------- state */

#define IDX3(s, x, y, z) ((size_t)(z) * (s)->ny * (s)->nx + (size_t)(y) * (s)->nx + (size_t)(x))

static float *zeros(size_t n) { return (float *)calloc(n ? n : 1, sizeof(float)); }

static orc_state *create_common(const orc_params *p, int mode, int ecoef, int gy0, int nyl)
{
    if (mode == ORC_MODE_2D_TMZ && p->nz != 1) return NULL;
    orc_state *s = (orc_state *)calloc(1, sizeof *s);
    s->p = *p;
    s->mode = mode;
    s->ecoef = ecoef;
    s->nx = p->nx;
    s->nz = p->nz;
    s->gny = p->ny;
    if (mode == ORC_MODE_2D_TMZ) {
        s->gy0 = gy0;
        s->nyl = nyl;
        s->ny = nyl + 2;
    } else {
        s->gy0 = 0;
        s->nyl = p->ny;
        s->ny = p->ny;
    }
    s->n = (size_t)s->nx * s->ny * s->nz;
    for (int k = 0; k < 3; k++) s->h[k] = zeros(s->n);
    for (int b = 0; b < 3; b++)
        for (int k = 0; k < 3; k++) {
            s->d[b][k] = zeros(s->n);
            s->e[b][k] = zeros(s->n);
        }
    s->sigma = zeros(s->n);
    s->eps_i = zeros(s->n);
    s->eps_s = zeros(s->n);
    s->tau_d = zeros(s->n);
    for (size_t i = 0; i < s->n; i++) {                        /* fdtd.cu:708-713 */
        s->sigma[i] = p->default_sigma;
        s->eps_s[i] = p->default_eps_s;
        s->eps_i[i] = p->default_eps_i;
        s->tau_d[i] = p->default_tau_d;
    }
    /* rp planes are always indexed by GLOBAL coordinates. */
    s->rpn[0] = s->rpn[3] = (size_t)2 * s->gny * s->nz;        /* fdtd.cu:716,720 */
    s->rpn[1] = s->rpn[4] = (size_t)s->nx * 2 * s->nz;         /* fdtd.cu:717,721 */
    s->rpn[2] = s->rpn[5] = (size_t)s->nx * s->gny * 2;        /* fdtd.cu:718,722 */
    for (int f = 0; f < 6; f++) s->rp[f] = zeros(s->rpn[f]);
    s->it = 0;
    return s;
}

/* Re{eps_i + (eps_s - eps_i)/(1 + c1) - c2}, fdtd.cu:959-963 (x faces: c1 and
 * c2 are real because the "* I" is missing) and :983-987 (y, z faces). */
static float rp_value(const orc_params *p, float sigma, float eps_s, float eps_i, float tau,
                      int x_face)
{
    float complex c1, c2;
    if (x_face) {
        c1 = 0.0 + 2.0 * p->pi * p->wave_frequency * tau;
        c2 = 0.0 + sigma / (2.0 * p->pi * p->wave_frequency * p->eps0);
    } else {
        c1 = 0.0 + 2.0 * p->pi * p->wave_frequency * tau * I;
        c2 = 0.0 + sigma / (2.0 * p->pi * p->wave_frequency * p->eps0) * I;
    }
    return creal(eps_i + (eps_s - eps_i) / (1.0 + c1) - c2);
}

/* Local storage index of global cell (x, gy, z), or -1 when the slab does not
 * hold that row. */
static long cell_of(const orc_state *s, int x, int gy, int z)
{
    if (s->mode == ORC_MODE_2D_TMZ) {
        int j = gy - s->gy0;
        if (j < -1 || j > s->nyl) return -1;
        return (long)(j + 1) * s->nx + x;
    }
    return (long)IDX3(s, x, gy, z);
}

/* fdtd.cu:942-1024 */
void orc_setup_mur(orc_state *s)
{
    const orc_params *p = &s->p;
    int nx = s->nx, ny = s->gny, nz = s->nz;
#define MAT(c) s->sigma[c], s->eps_s[c], s->eps_i[c], s->tau_d[c]
    for (int iz = 0; iz < nz; iz++)
        for (int iy = 0; iy < ny; iy++)
            for (int k = 0; k < 2; k++) {
                long c0 = cell_of(s, k, iy, iz), c1 = cell_of(s, nx - 2 + k, iy, iz);
                size_t r = (size_t)iz * ny * 2 + (size_t)iy * 2 + k;
                if (c0 >= 0) s->rp[0][r] = rp_value(p, MAT(c0), 1);
                if (c1 >= 0) s->rp[3][r] = rp_value(p, MAT(c1), 1);
            }
    for (int iz = 0; iz < nz; iz++)
        for (int ix = 0; ix < nx; ix++)
            for (int k = 0; k < 2; k++) {
                long c0 = cell_of(s, ix, k, iz), c1 = cell_of(s, ix, ny - 2 + k, iz);
                size_t r = (size_t)iz * 2 * nx + (size_t)k * nx + ix;
                if (c0 >= 0) s->rp[1][r] = rp_value(p, MAT(c0), 0);
                if (c1 >= 0) s->rp[4][r] = rp_value(p, MAT(c1), 0);
            }
    if (s->mode == ORC_MODE_3D)
        for (int iy = 0; iy < ny; iy++)
            for (int ix = 0; ix < nx; ix++)
                for (int k = 0; k < 2; k++) {
                    long c0 = cell_of(s, ix, iy, k), c1 = cell_of(s, ix, iy, nz - 2 + k);
                    size_t r = (size_t)k * ny * nx + (size_t)iy * nx + ix;
                    s->rp[2][r] = rp_value(p, MAT(c0), 0);
                    s->rp[5][r] = rp_value(p, MAT(c1), 0);
                }
#undef MAT
}

/* The reference calls setupMurBoundary while every cell still holds the
 * defaults (fdtd.cu:59 runs before :65), so each plane is one constant. */
static void mur_from_defaults(orc_state *s)
{
    const orc_params *p = &s->p;
    float rx = rp_value(p, p->default_sigma, p->default_eps_s, p->default_eps_i, p->default_tau_d, 1);
    float ryz = rp_value(p, p->default_sigma, p->default_eps_s, p->default_eps_i, p->default_tau_d, 0);
    for (int f = 0; f < 6; f++)
        for (size_t i = 0; i < s->rpn[f]; i++) s->rp[f][i] = (f % 3 == 0) ? rx : ryz;
}

orc_state *orc_create(const orc_params *p, int mode, int ecoef)
{
    if (p->nx < 3 || p->ny < 3 || (mode == ORC_MODE_3D && p->nz < 3)) return NULL;
    orc_state *s = create_common(p, mode, ecoef, 0, p->ny);
    if (s) mur_from_defaults(s);
    return s;
}

orc_state *orc_create_slab2d(const orc_params *p, int ecoef, int gy0, int nyl)
{
    /* a slab needs its own first interior row for the y-face Mur update */
    if (p->nx < 3 || p->ny < 3 || gy0 < 0 || nyl < 2 || gy0 + nyl > p->ny) return NULL;
    orc_state *s = create_common(p, ORC_MODE_2D_TMZ, ecoef, gy0, nyl);
    if (s) mur_from_defaults(s);
    return s;
}

void orc_destroy(orc_state *s)
{
    if (!s) return;
    for (int k = 0; k < 3; k++) free(s->h[k]);
    for (int b = 0; b < 3; b++)
        for (int k = 0; k < 3; k++) {
            free(s->d[b][k]);
            free(s->e[b][k]);
        }
    free(s->sigma);
    free(s->eps_i);
    free(s->eps_s);
    free(s->tau_d);
    for (int f = 0; f < 6; f++) free(s->rp[f]);
    free(s);
}

int orc_set_rp(orc_state *s, int face, const float *data, size_t n)
{
    if (face < 0 || face > 5 || n != s->rpn[face]) return -1;
    memcpy(s->rp[face], data, n * sizeof(float));
    return 0;
}

float *orc_rp(orc_state *s, int face, size_t *n)
{
    if (face < 0 || face > 5) return NULL;
    if (n) *n = s->rpn[face];
    return s->rp[face];
}

/* Second half of loadMaterials, fdtd.cu:922-931, on caller data.  codes is in
 * storage order (2-D slab: nyl + 2 rows, halo rows included). */
int orc_set_materials(orc_state *s, const uint8_t *codes, const float *table4, int ncodes)
{
    for (size_t i = 0; i < s->n; i++) {
        int code = codes[i];
        if (code >= ncodes) return -1;
        s->sigma[i] = table4[code * 4 + 0];
        s->eps_s[i] = table4[code * 4 + 1];
        s->eps_i[i] = table4[code * 4 + 2];
        s->tau_d[i] = table4[code * 4 + 3];
    }
    return 0;
}

/* fdtd.cu:875-939.  Differences from the reference, both forced: the table
 * has 256 rows (the reference's 94-row table is overrun by code 94, which the
 * shipped data uses), and running out of tokens is an error (the reference
 * ignores fscanf failures). */
int orc_load_materials(orc_state *s, const char *specs_path, const char *pgm_dir, int first_slice)
{
    float *specs = (float *)calloc((size_t)ORC_MAX_CODES * 4, sizeof(float));
    cursor c;
    char name[1024];
    if (slurp(specs_path, &c) != 0) { free(specs); return -1; }
    for (int i = 0; i < 94; i++) {                             /* :894-904 */
        int index;
        float v[4];
        if (scan_int(&c, &index) || !token(&c, name, sizeof name) || scan_float(&c, &v[0]) ||
            scan_float(&c, &v[1]) || scan_float(&c, &v[2]) || scan_float(&c, &v[3]))
            break;
        if (index < 0 || index >= ORC_MAX_CODES) { free(c.buf); free(specs); return -2; }
        memcpy(&specs[index * 4], v, sizeof v);
        if (index >= 94) break;
    }
    free(c.buf);

    int rc = 0;
    int zcount = (s->mode == ORC_MODE_2D_TMZ) ? 1 : s->nz;
    for (int iz = 0; iz < zcount && rc == 0; iz++) {           /* :909-936 */
        char path[2048];
        snprintf(path, sizeof path, "%s/v1_%05d.pgm", pgm_dir, iz + first_slice);
        if (slurp(path, &c) != 0) { rc = -3; break; }
        int w, h;
        if (!token(&c, name, sizeof name) || !token(&c, name, sizeof name) ||
            !token(&c, name, sizeof name) || scan_int(&c, &w) || scan_int(&c, &h) ||
            !token(&c, name, sizeof name)) {                   /* :920 */
            rc = -4;
        }
        /* nx*ny tokens in sequence, x fastest; the declared size is ignored. */
        for (int iy = 0; iy < s->gny && rc == 0; iy++)
            for (int ix = 0; ix < s->nx; ix++) {
                int code;
                if (scan_int(&c, &code)) { rc = -5; break; }
                if (code < 0 || code >= ORC_MAX_CODES) { rc = -6; break; }
                long o = cell_of(s, ix, iy, iz);
                if (o < 0) continue;
                s->sigma[o] = specs[code * 4 + 0];
                s->eps_s[o] = specs[code * 4 + 1];
                s->eps_i[o] = specs[code * 4 + 2];
                s->tau_d[o] = specs[code * 4 + 3];
            }
        free(c.buf);
    }
    free(specs);
    return rc;
}

/* ---------------------------------------------------------- field access */

static int buffer_of_level(const orc_state *s, int level)
{
    return (((s->it - 1 - level) % 3) + 3) % 3;
}

float *orc_field_level(orc_state *s, int field_id, int level)
{
    if (field_id < 0 || field_id >= ORC_NFIELDS || level < 0 || level > 2) return NULL;
    int k = field_id % 3;
    if (field_id <= ORC_HZ) return s->h[k];
    int b = buffer_of_level(s, level);
    return (field_id <= ORC_DZ) ? s->d[b][k] : s->e[b][k];
}

float *orc_field(orc_state *s, int field_id) { return orc_field_level(s, field_id, 0); }
size_t orc_field_count(const orc_state *s) { return s->n; }
int orc_steps_done(const orc_state *s) { return s->it; }

/* ------------------------------------------------------------ arithmetic */

typedef struct { float a, b, c, d; } ecoefs;

/* fdtd_calculations.cu:141-144; CONSISTENT replaces 2b by a+b in d. */
static inline ecoefs e_coefs(const orc_state *s, size_t i)
{
    const float dEps0 = s->p.eps0, dDt = s->p.dt;
    const float epsI = s->eps_i[i], tauD = s->tau_d[i], epsS = s->eps_s[i], sigma = s->sigma[i];
    ecoefs k;
    k.a = 2.0 * dEps0 * epsI * tauD;
    k.b = k.a + 2.0 * dDt * (dEps0 * epsS + sigma * tauD);
    k.c = 1.0 / (k.b + sigma * dDt * dDt);
    if (s->ecoef == ORC_ECOEF_REFERENCE)
        k.d = (2.0 * k.b - sigma * dDt * dDt);
    else
        k.d = ((double)k.a + (double)k.b - sigma * dDt * dDt);
    return k;
}

void orc_cell_coefs(const orc_state *s, size_t cell, float abcd[4])
{
    ecoefs k = e_coefs(s, cell);
    abcd[0] = k.a;
    abcd[1] = k.b;
    abcd[2] = k.c;
    abcd[3] = k.d;
}

float *orc_material(orc_state *s, int which)
{
    switch (which) {
    case 0: return s->sigma;
    case 1: return s->eps_s;
    case 2: return s->eps_i;
    case 3: return s->tau_d;
    default: return NULL;
    }
}

/* fdtd_calculations.cu:150-156 */
static inline float e_update(const orc_state *s, size_t i, ecoefs k, float e0, float e1, float d0,
                             float d1, float d2)
{
    const float dDt = s->p.dt, tauD = s->tau_d[i];
    return k.c * (k.d * e0 - k.a * e1 + (2.0 * (dDt + tauD)) * d0 -
                  (2.0 * dDt + 4.0 * tauD) * d1 + (2.0 * tauD) * d2);
}

/* fdtd_calculations.cu:240-248 and its eleven siblings: r_here is rp at the
 * face layer, r_in at the first interior layer. */
static inline float mur_face(const orc_state *s, float du, float r_here, float r_in, float et_in,
                             float es_in, float es_here)
{
    const float dDt = s->p.dt, dMu0 = s->p.mu0, dEps0 = s->p.eps0;
    return 1 / (dDt + du * sqrtf(dMu0 * dEps0 * r_here)) *
           ((dDt - du * sqrtf(dMu0 * dEps0 * r_in)) * et_in +
            (dDt + du * sqrtf(dMu0 * dEps0 * r_in)) * es_in -
            (dDt - du * sqrtf(dMu0 * dEps0 * r_here)) * es_here);
}

static inline float jz_at(const orc_state *s, int i)
{
    /* fdtd_calculations.cu:27-28 uploads min(512, jzCount) samples; the
     * reference reads past them for i >= 512, here that is 0. */
    return (i >= 0 && i < ORC_MAX_JZ) ? s->p.jz[i] : 0.0f;
}

/* ------------------------------------------------------------- 3-D step */

#define OMP_FOR _Pragma("omp parallel for collapse(2) schedule(static) num_threads(threads)")

/* fdtd_calculations.cu:32-77 */
static void h3(orc_state *s, float *const e[3], int threads)
{
    const int nx = s->nx, ny = s->ny, nz = s->nz;
    const float dDt = s->p.dt, dMu0 = s->p.mu0, dDx = s->p.dx, dDy = s->p.dy, dDz = s->p.dz;
    float *hx = s->h[0], *hy = s->h[1], *hz = s->h[2];
    const float *ex = e[0], *ey = e[1], *ez = e[2];
    OMP_FOR
    for (int iz = 0; iz < nz; iz++)
        for (int iy = 0; iy < ny; iy++)
            for (int ix = 0; ix < nx; ix++) {
                size_t o = IDX3(s, ix, iy, iz);
                if (ix >= 1 && ix < nx - 1 && iy >= 0 && iy < ny - 1 && iz >= 0 && iz < nz - 1)
                    hx[o] = hx[o] - dDt / (dMu0 * dDy) * (ez[IDX3(s, ix, iy + 1, iz)] - ez[o]) +
                            dDt / (dMu0 * dDy) * (ey[IDX3(s, ix, iy, iz + 1)] - ey[o]);
                if (ix >= 0 && ix < nx - 1 && iy >= 1 && iy < ny - 1 && iz >= 0 && iz < nz - 1)
                    hy[o] = hy[o] - dDt / (dMu0 * dDz) * (ex[IDX3(s, ix, iy, iz + 1)] - ex[o]) +
                            dDt / (dMu0 * dDx) * (ez[IDX3(s, ix + 1, iy, iz)] - ez[o]);
                if (ix >= 0 && ix < nx - 1 && iy >= 0 && iy < ny - 1 && iz >= 1 && iz < nz - 1)
                    hz[o] = hz[o] - dDt / (dMu0 * dDx) * (ey[IDX3(s, ix + 1, iy, iz)] - ey[o]) +
                            dDt / (dMu0 * dDy) * (ex[IDX3(s, ix, iy + 1, iz)] - ex[o]);
            }
}

/* fdtd_calculations.cu:80-122 */
static void d3(orc_state *s, float *const dt_[3], float *const ds[3], int threads)
{
    const int nx = s->nx, ny = s->ny, nz = s->nz;
    const float x = s->p.dt / s->p.dx, y = s->p.dt / s->p.dy, z = s->p.dt / s->p.dz;
    const float *hx = s->h[0], *hy = s->h[1], *hz = s->h[2];
    OMP_FOR
    for (int iz = 0; iz < nz; iz++)
        for (int iy = 0; iy < ny; iy++)
            for (int ix = 0; ix < nx; ix++) {
                size_t o = IDX3(s, ix, iy, iz);
                if (ix >= 0 && ix < nx - 1 && iy >= 1 && iy < ny - 1 && iz >= 1 && iz < nz - 1)
                    dt_[0][o] = ds[0][o] + y * (hz[o] - hz[IDX3(s, ix, iy - 1, iz)]) -
                                z * (hy[o] - hy[IDX3(s, ix, iy, iz - 1)]);
                if (ix >= 1 && ix < nx - 1 && iy >= 0 && iy < ny - 1 && iz >= 1 && iz < nz - 1)
                    dt_[1][o] = ds[1][o] + z * (hx[o] - hx[IDX3(s, ix, iy, iz - 1)]) -
                                x * (hz[o] - hz[IDX3(s, ix - 1, iy, iz)]);
                if (ix >= 1 && ix < nx - 1 && iy >= 1 && iy < ny - 1 && iz >= 0 && iz < nz - 1)
                    dt_[2][o] = ds[2][o] + x * (hy[o] - hy[IDX3(s, ix - 1, iy, iz)]) -
                                y * (hx[o] - hx[IDX3(s, ix, iy - 1, iz)]);
            }
}

/* fdtd_calculations.cu:187-213 */
static void src3(orc_state *s, float *dzt, const float *dzs, int curr_iteration)
{
    const float dDt = s->p.dt, dDx = s->p.dx, dDy = s->p.dy;
    const float *hx = s->h[0], *hy = s->h[1];
    for (int i = 0; i < s->p.sources_count; i++) {
        int x = s->p.sources[i * 3 + 0], y = s->p.sources[i * 3 + 1], z = s->p.sources[i * 3 + 2];
        size_t o = IDX3(s, x, y, z);
        float val = dzs[o] + dDt / dDx * (hy[o] - hy[IDX3(s, x - 1, y, z)]) -
                    dDt / dDy * (hx[o] - hx[IDX3(s, x, y - 1, z)]) - jz_at(s, curr_iteration);
        dzt[o] = val;
    }
}

/* fdtd_calculations.cu:125-184 */
static void e3(orc_state *s, float *const et[3], float *const e0[3], float *const e1[3],
               float *const d0[3], float *const d1[3], float *const d2[3], int threads)
{
    const int nx = s->nx, ny = s->ny, nz = s->nz;
    OMP_FOR
    for (int iz = 0; iz < nz; iz++)
        for (int iy = 0; iy < ny; iy++)
            for (int ix = 0; ix < nx; ix++) {
                size_t o = IDX3(s, ix, iy, iz);
                ecoefs k = e_coefs(s, o);
                if (ix >= 0 && ix < nx - 1 && iy >= 1 && iy < ny - 1 && iz >= 1 && iz < nz - 1)
                    et[0][o] = e_update(s, o, k, e0[0][o], e1[0][o], d0[0][o], d1[0][o], d2[0][o]);
                if (ix >= 1 && ix <= nx - 1 && iy >= 0 && iy <= ny - 1 && iz >= 1 && iz <= nz - 1) {
                    et[1][o] = e_update(s, o, k, e0[1][o], e1[1][o], d0[1][o], d1[1][o], d2[1][o]);
                    et[2][o] = e_update(s, o, k, e0[2][o], e1[2][o], d0[2][o], d1[2][o], d2[2][o]);
                }
            }
}

/* fdtd_calculations.cu:216-428 */
static void mur3(orc_state *s, float *const et[3], float *const es[3])
{
    const int nx = s->nx, ny = s->ny, nz = s->nz;
    const float dDx = s->p.dx, dDy = s->p.dy, dDz = s->p.dz;
    const float *rpx0 = s->rp[0], *rpy0 = s->rp[1], *rpz0 = s->rp[2];
    const float *rpxE = s->rp[3], *rpyE = s->rp[4], *rpzE = s->rp[5];
#define RPX(p, k, y, z) p[(size_t)(z) * ny * 2 + (size_t)(y) * 2 + (k)]
#define RPY(p, x, k, z) p[(size_t)(z) * 2 * nx + (size_t)(k) * nx + (x)]
#define RPZ(p, x, y, k) p[(size_t)(k) * ny * nx + (size_t)(y) * nx + (x)]
    float *ext = et[0], *eyt = et[1], *ezt = et[2];
    const float *exs = es[0], *eys = es[1], *ezs = es[2];

    /* ex on the y faces, :237-263 */
    for (int iz = 1; iz < nz - 1; iz++)
        for (int ix = 0; ix < nx - 1; ix++) {
            size_t f = IDX3(s, ix, 0, iz), in = IDX3(s, ix, 1, iz);
            ext[f] = mur_face(s, dDy, RPY(rpy0, ix, 0, iz), RPY(rpy0, ix, 1, iz), ext[in], exs[in], exs[f]);
            f = IDX3(s, ix, ny - 1, iz), in = IDX3(s, ix, ny - 2, iz);
            ext[f] = mur_face(s, dDy, RPY(rpyE, ix, 1, iz), RPY(rpyE, ix, 0, iz), ext[in], exs[in], exs[f]);
        }
    /* ex on the z faces, :269-295 */
    for (int iy = 1; iy < ny - 1; iy++)
        for (int ix = 0; ix < nx - 1; ix++) {
            size_t f = IDX3(s, ix, iy, 0), in = IDX3(s, ix, iy, 1);
            ext[f] = mur_face(s, dDz, RPZ(rpz0, ix, iy, 0), RPZ(rpz0, ix, iy, 1), ext[in], exs[in], exs[f]);
            f = IDX3(s, ix, iy, nz - 1), in = IDX3(s, ix, iy, nz - 2);
            ext[f] = mur_face(s, dDz, RPZ(rpzE, ix, iy, 1), RPZ(rpzE, ix, iy, 0), ext[in], exs[in], exs[f]);
        }
    /* ey on the x faces, :303-329.  The high face takes eySource for BOTH
     * interior terms (:323,:325) -- kept. */
    for (int iz = 1; iz < nz - 1; iz++)
        for (int iy = 0; iy < ny - 1; iy++) {
            size_t f = IDX3(s, 0, iy, iz), in = IDX3(s, 1, iy, iz);
            eyt[f] = mur_face(s, dDx, RPX(rpx0, 0, iy, iz), RPX(rpx0, 1, iy, iz), eyt[in], eys[in], eys[f]);
            f = IDX3(s, nx - 1, iy, iz), in = IDX3(s, nx - 2, iy, iz);
            eyt[f] = mur_face(s, dDx, RPX(rpxE, 1, iy, iz), RPX(rpxE, 0, iy, iz), eys[in], eys[in], eys[f]);
        }
    /* ey on the z faces, :335-361 */
    for (int iy = 0; iy < ny - 1; iy++)
        for (int ix = 1; ix < nx - 1; ix++) {
            size_t f = IDX3(s, ix, iy, 0), in = IDX3(s, ix, iy, 1);
            eyt[f] = mur_face(s, dDz, RPZ(rpz0, ix, iy, 0), RPZ(rpz0, ix, iy, 1), eyt[in], eys[in], eys[f]);
            f = IDX3(s, ix, iy, nz - 1), in = IDX3(s, ix, iy, nz - 2);
            eyt[f] = mur_face(s, dDz, RPZ(rpzE, ix, iy, 1), RPZ(rpzE, ix, iy, 0), eyt[in], eys[in], eys[f]);
        }
    /* ez on the x faces, :369-395 */
    for (int iz = 0; iz < nz - 1; iz++)
        for (int iy = 1; iy < ny - 1; iy++) {
            size_t f = IDX3(s, 0, iy, iz), in = IDX3(s, 1, iy, iz);
            ezt[f] = mur_face(s, dDx, RPX(rpx0, 0, iy, iz), RPX(rpx0, 1, iy, iz), ezt[in], ezs[in], ezs[f]);
            f = IDX3(s, nx - 1, iy, iz), in = IDX3(s, nx - 2, iy, iz);
            ezt[f] = mur_face(s, dDx, RPX(rpxE, 1, iy, iz), RPX(rpxE, 0, iy, iz), ezt[in], ezs[in], ezs[f]);
        }
    /* ez on the y faces, :401-427 */
    for (int iz = 0; iz < nz - 1; iz++)
        for (int ix = 1; ix < nx - 1; ix++) {
            size_t f = IDX3(s, ix, 0, iz), in = IDX3(s, ix, 1, iz);
            ezt[f] = mur_face(s, dDy, RPY(rpy0, ix, 0, iz), RPY(rpy0, ix, 1, iz), ezt[in], ezs[in], ezs[f]);
            f = IDX3(s, ix, ny - 1, iz), in = IDX3(s, ix, ny - 2, iz);
            ezt[f] = mur_face(s, dDy, RPY(rpyE, ix, 1, iz), RPY(rpyE, ix, 0, iz), ezt[in], ezs[in], ezs[f]);
        }
}

/* ------------------------------------------------------- 2-D TMz step
 *
 * The reference has no 2-D mode: with nz = 1 every z guard fails.  The 2-D
 * TMz reduction keeps hx, hy, dz, ez and drops every z difference and every
 * z guard from the 3-D formulas; dt keeps the 3-D Courant factor.  Storage
 * row r holds global row gy0 + r - 1; rows 0 and nyl+1 are halo rows that a
 * slab driver refreshes with the neighbour's ez (and that are zero outside
 * the grid).  The hx halo row below the slab is recomputed here each step
 * from the ez halo, so only ez ever has to be exchanged. */

#define IDX2(s, x, j) ((size_t)((j) + 1) * (s)->nx + (size_t)(x))

static void h2(orc_state *s, const float *ez, int threads)
{
    const int nx = s->nx, gny = s->gny;
    const float dDt = s->p.dt, dMu0 = s->p.mu0, dDx = s->p.dx, dDy = s->p.dy;
    float *hx = s->h[0], *hy = s->h[1];
    const int jlo = (s->gy0 > 0) ? -1 : 0;
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int j = jlo; j < s->nyl; j++) {
        const int iy = s->gy0 + j;
        for (int ix = 0; ix < nx; ix++) {
            size_t o = IDX2(s, ix, j);
            if (ix >= 1 && ix < nx - 1 && iy >= 0 && iy < gny - 1)
                hx[o] = hx[o] - dDt / (dMu0 * dDy) * (ez[IDX2(s, ix, j + 1)] - ez[o]);
            if (j >= 0 && ix >= 0 && ix < nx - 1 && iy >= 1 && iy < gny - 1)
                hy[o] = hy[o] + dDt / (dMu0 * dDx) * (ez[o + 1] - ez[o]);
        }
    }
}

static void d2(orc_state *s, float *dzt, const float *dzs, int curr_iteration, int threads)
{
    const int nx = s->nx, gny = s->gny;
    const float x = s->p.dt / s->p.dx, y = s->p.dt / s->p.dy;
    const float *hx = s->h[0], *hy = s->h[1];
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int j = 0; j < s->nyl; j++) {
        const int iy = s->gy0 + j;
        for (int ix = 0; ix < nx; ix++) {
            size_t o = IDX2(s, ix, j);
            if (ix >= 1 && ix < nx - 1 && iy >= 1 && iy < gny - 1)
                dzt[o] = dzs[o] + x * (hy[o] - hy[o - 1]) - y * (hx[o] - hx[IDX2(s, ix, j - 1)]);
        }
    }
    /* sources, fdtd_calculations.cu:199-211 without the z coordinate */
    const float dDt = s->p.dt, dDx = s->p.dx, dDy = s->p.dy;
    for (int i = 0; i < s->p.sources_count; i++) {
        int sx = s->p.sources[i * 3 + 0], sy = s->p.sources[i * 3 + 1];
        int j = sy - s->gy0;
        if (j < 0 || j >= s->nyl) continue;
        size_t o = IDX2(s, sx, j);
        float val = dzs[o] + dDt / dDx * (hy[o] - hy[o - 1]) -
                    dDt / dDy * (hx[o] - hx[IDX2(s, sx, j - 1)]) - jz_at(s, curr_iteration);
        dzt[o] = val;
    }
}

static void e2(orc_state *s, float *et, const float *e0, const float *e1, const float *d0,
               const float *d1, const float *d2_, int threads)
{
    const int nx = s->nx;
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int j = 0; j < s->nyl; j++)
        for (int ix = 1; ix <= nx - 1; ix++) {      /* ez guard, fdtd_calculations.cu:173-175 */
            size_t o = IDX2(s, ix, j);
            ecoefs k = e_coefs(s, o);
            et[o] = e_update(s, o, k, e0[o], e1[o], d0[o], d1[o], d2_[o]);
        }
}

static void mur2(orc_state *s, float *ezt, const float *ezs)
{
    const int nx = s->nx, gny = s->gny;
    const float dDx = s->p.dx, dDy = s->p.dy;
    const float *rpx0 = s->rp[0], *rpy0 = s->rp[1], *rpxE = s->rp[3], *rpyE = s->rp[4];
    /* ez on the x faces, fdtd_calculations.cu:369-395 with iz = 0 */
    for (int j = 0; j < s->nyl; j++) {
        const int iy = s->gy0 + j;
        if (iy < 1 || iy > gny - 2) continue;
        size_t f = IDX2(s, 0, j), in = IDX2(s, 1, j);
        ezt[f] = mur_face(s, dDx, rpx0[iy * 2 + 0], rpx0[iy * 2 + 1], ezt[in], ezs[in], ezs[f]);
        f = IDX2(s, nx - 1, j), in = IDX2(s, nx - 2, j);
        ezt[f] = mur_face(s, dDx, rpxE[iy * 2 + 1], rpxE[iy * 2 + 0], ezt[in], ezs[in], ezs[f]);
    }
    /* ez on the y faces, fdtd_calculations.cu:401-427 with iz = 0 */
    if (s->gy0 == 0)
        for (int ix = 1; ix < nx - 1; ix++) {
            size_t f = IDX2(s, ix, 0), in = IDX2(s, ix, 1);
            ezt[f] = mur_face(s, dDy, rpy0[0 * nx + ix], rpy0[1 * nx + ix], ezt[in], ezs[in], ezs[f]);
        }
    if (s->gy0 + s->nyl == gny)
        for (int ix = 1; ix < nx - 1; ix++) {
            int j = s->nyl - 1;
            size_t f = IDX2(s, ix, j), in = IDX2(s, ix, j - 1);
            ezt[f] = mur_face(s, dDy, rpyE[1 * nx + ix], rpyE[0 * nx + ix], ezt[in], ezs[in], ezs[f]);
        }
}

/* ----------------------------------------------------------------- step */

/* fdtd.cu:115-530: one pass of the hand-unrolled loop body.  Sub-step `it`
 * writes buffer t = it % 3, reads s = (t+2) % 3 as time n and o = (t+1) % 3 as
 * time n-1 (:206-211, :344-349, :482-487); all three sub-steps of a trip pass
 * the trip's first index to updateSources (:172, :310, :448). */
static void step_once(orc_state *s, int threads)
{
    const int it = s->it;
    const int t = it % 3, sN = (t + 2) % 3, o = (t + 1) % 3;
    const int curr_iteration = it - it % 3;
    if (s->mode == ORC_MODE_3D) {
        h3(s, s->e[sN], threads);
        d3(s, s->d[t], s->d[sN], threads);
        src3(s, s->d[t][2], s->d[sN][2], curr_iteration);
        e3(s, s->e[t], s->e[sN], s->e[o], s->d[t], s->d[sN], s->d[o], threads);
        mur3(s, s->e[t], s->e[sN]);
    } else {
        h2(s, s->e[sN][2], threads);
        d2(s, s->d[t][2], s->d[sN][2], curr_iteration, threads);
        e2(s, s->e[t][2], s->e[sN][2], s->e[o][2], s->d[t][2], s->d[sN][2], s->d[o][2], threads);
        mur2(s, s->e[t][2], s->e[sN][2]);
    }
    s->it = it + 1;
}

void orc_step(orc_state *s, int nsteps, int threads)
{
    if (threads < 1) threads = 1;
    for (int i = 0; i < nsteps; i++) step_once(s, threads);
}

double orc_time_steps(orc_state *s, int nsteps, int threads)
{
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    orc_step(s, nsteps, threads);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
