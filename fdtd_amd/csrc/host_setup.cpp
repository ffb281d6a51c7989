// Host-side input surface of libfdtd_amd.so: the data/input_params parser, the
// source waveform, the material specs + PGM geometry readers, the E-update
// coefficient table and the Mur coefficient values.  Each function names the
// reference code (UnalignedByte/fdtd) whose results it reproduces; float and
// double are kept apart exactly where the reference's C promotions put them.
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <sstream>

#include "internal.h"

namespace fdtd {

namespace {

thread_local std::string g_error;

bool read_file(const char *path, std::string &out)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    out.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    return true;
}

// Reader with scanf-like primitives over a whole file held in memory.
class Scanner {
  public:
    explicit Scanner(const std::string &text) : s_(text), i_(0) {}

    void skip_space()
    {
        while (i_ < s_.size() && std::isspace((unsigned char)s_[i_])) ++i_;
    }
    // "%s"
    bool word(std::string *out = nullptr)
    {
        skip_space();
        if (i_ >= s_.size()) return false;
        size_t b = i_;
        while (i_ < s_.size() && !std::isspace((unsigned char)s_[i_])) ++i_;
        if (out) out->assign(s_, b, i_ - b);
        return true;
    }
    // "%d"
    bool integer(int *v)
    {
        skip_space();
        const char *b = s_.c_str() + i_;
        char *e = nullptr;
        long x = std::strtol(b, &e, 10);
        if (e == b) return false;
        i_ += (size_t)(e - b);
        *v = (int)x;
        return true;
    }
    // "%g" / "%f"
    bool real(float *v)
    {
        skip_space();
        const char *b = s_.c_str() + i_;
        char *e = nullptr;
        float x = std::strtof(b, &e);
        if (e == b) return false;
        i_ += (size_t)(e - b);
        *v = x;
        return true;
    }
    // fgets(): drop the rest of the current line
    void rest_of_line()
    {
        while (i_ < s_.size() && s_[i_] != '\n') ++i_;
        if (i_ < s_.size()) ++i_;
    }

  private:
    const std::string &s_;
    size_t i_;
};

} // namespace

void set_error(const std::string &msg) { g_error = msg; }

} // namespace fdtd

extern "C" const char *fdtd_last_error(void) { return fdtd::g_error.c_str(); }

namespace fdtd {

// fdtd.cu:625-634
void params_derive(fdtd_params *p)
{
    p->pi = (float)std::acos(-1.0);
    p->c = (float)(3.0 * std::pow(10.0, 8.0));
    p->timeskip = 1.0f;
    p->lambda = p->c / p->wave_frequency;
    p->dx = p->lambda / (float)p->elements_per_wave;
    p->dy = p->dx;
    p->dz = p->dx;
    const double inv2 = 1.0 / std::pow((double)p->dx, 2.0) + 1.0 / std::pow((double)p->dy, 2.0) +
                        1.0 / std::pow((double)p->dz, 2.0);
    p->dt = (float)(1.0 * (double)p->timeskip / ((double)p->c * std::sqrt(inv2)));
    p->mu0 = (float)(4.0 * (double)p->pi * std::pow(10.0, -7.0));
    const float mcc = (p->mu0 * p->c) * p->c;
    p->eps0 = (float)(1.0 / (double)mcc);
}

// fdtd.cu:547-623.  Positional: keys are read and ignored; a value line is
// "%s <values>\n" (the trailing \n eats following blank space), an unused line
// is dropped whole.
int params_load(const char *path, fdtd_params *out)
{
    std::string text;
    if (!read_file(path, text)) {
        set_error(std::string("cannot open ") + path);
        return -1;
    }
    fdtd_params p;
    std::memset(&p, 0, sizeof p);
    Scanner in(text);
    std::string tmp;
    auto fail = [&](const char *what) {
        set_error(std::string(path) + ": cannot read " + what);
        return -2;
    };

    if (!in.word() || !in.integer(&p.nx) || !in.integer(&p.ny) || !in.integer(&p.nz)) return fail("nx_ny_nz");
    in.skip_space();
    if (!in.word() || !in.integer(&p.iterations)) return fail("t_max");
    in.skip_space();
    p.iterations = ((p.iterations - 1) / 3 + 1) * 3; // fdtd.cu:563
    in.rest_of_line();                               // nf
    if (!in.word() || !in.word(&tmp)) return fail("env_set_dir");
    std::snprintf(p.input_path, sizeof p.input_path, "%s", tmp.c_str());
    in.skip_space();
    in.rest_of_line(); // env_file_prefix
    if (!in.word() || !in.word(&tmp)) return fail("output_dir");
    std::snprintf(p.output_path, sizeof p.output_path, "%s", tmp.c_str());
    in.skip_space();
    // output_format, impulse_resp_flag, pec_flag, read_env_flag, output_flag,
    // bzip2_flag, output_start, output_finish, source_type
    for (int i = 0; i < 9; ++i) in.rest_of_line();
    if (!in.word() || !in.integer(&p.elements_per_wave)) return fail("elements_per_wavelength");
    in.skip_space();
    if (!in.word() || !in.real(&p.wave_frequency)) return fail("wave_frequency");
    in.skip_space();
    if (!in.word() || !in.real(&p.pulse_width)) return fail("pulse_width");
    in.skip_space();
    if (!in.word() || !in.real(&p.pulse_modulation_frequency)) return fail("pulse_modulation_frequency");
    in.skip_space();
    if (!in.word() || !in.integer(&p.sources_count)) return fail("number_of_excitation_sources");
    in.skip_space();
    if (p.sources_count < 0 || p.sources_count > FDTD_MAX_SOURCES) {
        set_error("number_of_excitation_sources must be 0.." + std::to_string(FDTD_MAX_SOURCES));
        return -3;
    }
    for (int i = 0; i < p.sources_count; ++i) {
        if (!in.word()) return fail("source_location");
        for (int k = 0; k < 3; ++k) {
            int v;
            if (!in.integer(&v)) return fail("source_location");
            p.sources[i * 3 + k] = v - 1; // 1-based in the file, fdtd.cu:607-609
        }
        in.skip_space();
    }
    in.rest_of_line(); // pulse_type
    if (!in.word() || !in.real(&p.default_sigma)) return fail("fsigma");
    if (!in.word() || !in.real(&p.default_eps_s)) return fail("feps_s");
    if (!in.word() || !in.real(&p.default_eps_i)) return fail("feps_inf");
    if (!in.word() || !in.real(&p.default_tau_d)) return fail("ftau_d");
    if (p.elements_per_wave <= 0 || !(p.wave_frequency > 0.0f)) {
        set_error("elements_per_wavelength and wave_frequency must be positive");
        return -3;
    }
    params_derive(&p);
    *out = p;
    return 0;
}

// fdtd.cu:1027-1082: Gaussian (optionally cosine-modulated) pulse on a grid
// `fine` times finer than dt, decimated, scaled, then centrally differenced.
int setup_sources(fdtd_params *p)
{
    const float pw = p->pulse_width, f = p->wave_frequency, dt = p->dt;
    const int span = 1 << 16;
    const int fine = (int)((((float)(1 << 13) * pw) * f) * dt);
    if (fine <= 0) {
        set_error("setup_sources: 8192*pulse_width*wave_frequency*dt truncates to 0");
        return -4;
    }
    const float dt_fine = dt / (float)fine;
    const int half = (int)(1.0 / (double)(pw * f) / (double)dt_fine / 2.0);
    if (half + 1 >= span) {
        set_error("setup_sources: pulse longer than the waveform buffer");
        return -4;
    }
    std::vector<float> pulse((size_t)span * 2, 0.0f);
    const double quarter = ((double)(float)half + 1.0) / 4.0;
    const double two_pi = 2.0 * std::acos(-1.0);
    for (int i = -half - 1; i <= half + 1; ++i) {
        const float u = (float)((double)(float)i / quarter);
        const float env = (float)std::exp(-std::pow((double)u, 2.0));
        const double phase = two_pi * (double)p->pulse_modulation_frequency * (double)f * (double)i * (double)dt_fine;
        const float mod = (float)std::cos(phase);
        pulse[(size_t)(i + span)] = env * mod;
    }
    int first = 0;
    bool found = false;
    for (int i = -(1 << 12); i < (1 << 12) && !found; ++i) {
        if ((double)std::fabs(pulse[(size_t)(i + span)]) >= 1e-9 && i % fine == 0) {
            first = i;
            found = true;
        }
    }
    if (!found) {
        set_error("setup_sources: no pulse sample reaches 1e-9");
        return -4;
    }
    std::vector<float> coarse((size_t)span, 0.0f);
    size_t n = 0;
    for (int i = first; i <= half + 1; i += fine)
        coarse[n++] = (float)((double)pulse[(size_t)(i + span)] * 1e-15 / (double)dt / 3.0);

    const double area = (double)(p->dt * p->dz);
    const double volume = (double)((p->dx * p->dy) * p->dz);
    for (int k = 0; k < FDTD_MAX_JZ; ++k) {
        const float slope = ((coarse[k + 2] - coarse[k + 1]) / dt) + ((coarse[k + 1] - coarse[k]) / dt);
        p->jz[k] = (float)((double)slope / 2.0 * area / volume);
    }
    p->jz_count = FDTD_MAX_JZ;
    return 0;
}

// fdtd_calculations.cu:141-156 per material instead of per cell.
MatCoef make_coef(const MaterialRow &m, float dt, float eps0, int ecoef)
{
    MatCoef k;
    std::memset(&k, 0, sizeof k);
    k.a = (float)(2.0 * (double)eps0 * (double)m.eps_inf * (double)m.tau_d);
    const float inner = eps0 * m.eps_s + m.sigma * m.tau_d;
    const float b = (float)((double)k.a + 2.0 * (double)dt * (double)inner);
    const float sdt2 = (m.sigma * dt) * dt;
    k.c = (float)(1.0 / (double)(b + sdt2));
    if (ecoef == FDTD_ECOEF_REFERENCE)
        k.d = (float)(2.0 * (double)b - (double)sdt2);
    else
        k.d = (float)(((double)k.a + (double)b) - (double)sdt2);
    k.k1 = 2.0 * (double)(dt + m.tau_d);
    k.k2 = 2.0 * (double)dt + 4.0 * (double)m.tau_d;
    k.k3 = 2.0 * (double)m.tau_d;
    return k;
}

// fdtd.cu:959-963 (x faces) and :983-987 (y and z faces):
//   rp = Re{ eps_inf + (eps_s - eps_inf) / (1 + c1) - c2 }
// with c1 = 2*pi*f*tau and c2 = sigma / (2*pi*f*eps0) held as single-precision
// complex numbers; they are imaginary on the y and z faces and -- the "* I" is
// missing there -- real on the x faces.  The quotient real/complex follows the
// scaled (Smith) division a C compiler applies to double _Complex operands.
float rp_value(const fdtd_params &p, const MaterialRow &m, bool x_face)
{
    const double w = 2.0 * (double)p.pi * (double)p.wave_frequency;
    const float c1 = (float)(w * (double)m.tau_d);
    const float c2 = (float)((double)m.sigma / (w * (double)p.eps0));
    const double num = (double)(m.eps_s - m.eps_inf);
    double re;
    if (x_face) {
        re = num / (1.0 + (double)c1);
        return (float)(((double)m.eps_inf + re) - (double)c2);
    }
    const double c = 1.0, d = (double)c1;
    if (std::fabs(c) < std::fabs(d)) {
        const double ratio = c / d, denom = c * ratio + d;
        re = (num * ratio + 0.0) / denom;
    } else {
        const double ratio = d / c, denom = d * ratio + c;
        re = (0.0 * ratio + num) / denom;
    }
    return (float)((double)m.eps_inf + re);
}

// First half of loadMaterials, fdtd.cu:878-904: "<code> <name> <sigma> <eps_s>
// <eps_inf> <tau>" rows until 94 rows were read or a code >= 94 was seen.
int read_material_specs(const char *path, MaterialRow table[FDTD_MAX_CODES], bool present[FDTD_MAX_CODES])
{
    std::string text;
    if (!read_file(path, text)) {
        set_error(std::string("Couldn't open file ") + path);
        return -1;
    }
    Scanner in(text);
    for (int row = 0; row < 94; ++row) {
        int code;
        MaterialRow m;
        if (!in.integer(&code) || !in.word() || !in.real(&m.sigma) || !in.real(&m.eps_s) ||
            !in.real(&m.eps_inf) || !in.real(&m.tau_d))
            break;
        if (code < 0 || code >= kDefaultCode) {
            set_error(std::string(path) + ": material code " + std::to_string(code) + " out of range 0..254");
            return -2;
        }
        table[code] = m;
        present[code] = true;
        if (code >= 94) break;
    }
    return 0;
}

// Second half of loadMaterials, fdtd.cu:909-936, for one slice: skip the six
// header tokens (P2, #, <comment word>, width, height, maxval) and take the
// following integers in sequence, x fastest -- the declared width/height are
// not used to address the pixels.  With upsample u > 1 the sequence fills an
// (nx/u) x (ny/u) image whose pixels are repeated u x u times.
int read_pgm_base(const char *path, int bx, int by, uint8_t *base)
{
    std::string text;
    if (!read_file(path, text)) {
        set_error(std::string("Couldn't open file ") + path);
        return -1;
    }
    Scanner in(text);
    int w, h;
    if (!in.word() || !in.word() || !in.word() || !in.integer(&w) || !in.integer(&h) || !in.word()) {
        set_error(std::string(path) + ": bad PGM header");
        return -2;
    }
    const size_t n = (size_t)bx * by;
    for (size_t i = 0; i < n; ++i) {
        int code;
        if (!in.integer(&code)) {
            set_error(std::string(path) + ": fewer than nx*ny pixel values");
            return -2;
        }
        if (code < 0 || code >= kDefaultCode) {
            set_error(std::string(path) + ": material code " + std::to_string(code) + " out of range 0..254");
            return -2;
        }
        base[i] = (uint8_t)code;
    }
    return 0;
}

int read_pgm_slice(const char *path, int nx, int ny, int upsample, int y0, int rows, uint8_t *dst)
{
    if (upsample < 1 || nx % upsample || ny % upsample) {
        set_error("upsample must divide nx and ny");
        return -3;
    }
    const int bx = nx / upsample, by = ny / upsample;
    std::vector<uint8_t> base((size_t)bx * by);
    if (int rc = read_pgm_base(path, bx, by, base.data())) return rc;
    for (int r = 0; r < rows; ++r) {
        const uint8_t *src = &base[(size_t)((y0 + r) / upsample) * bx];
        uint8_t *row = dst + (size_t)r * nx;
        for (int x = 0; x < nx; ++x) row[x] = src[x / upsample];
    }
    return 0;
}

} // namespace fdtd

extern "C" {

int fdtd_params_derive(fdtd_params *p)
{
    if (!p || p->elements_per_wave <= 0 || !(p->wave_frequency > 0.0f)) {
        fdtd::set_error("fdtd_params_derive: elements_per_wave and wave_frequency must be positive");
        return -3;
    }
    fdtd::params_derive(p);
    return 0;
}

int fdtd_setup_sources(fdtd_params *p) { return fdtd::setup_sources(p); }

int fdtd_read_material_specs(const char *path, float *table4, uint8_t *present)
{
    fdtd::MaterialRow rows[FDTD_MAX_CODES];
    bool have[FDTD_MAX_CODES] = {};
    std::memset(rows, 0, sizeof rows);
    int rc = fdtd::read_material_specs(path, rows, have);
    if (rc) return rc;
    for (int i = 0; i < FDTD_MAX_CODES; ++i) {
        table4[i * 4 + 0] = rows[i].sigma;
        table4[i * 4 + 1] = rows[i].eps_s;
        table4[i * 4 + 2] = rows[i].eps_inf;
        table4[i * 4 + 3] = rows[i].tau_d;
        present[i] = have[i] ? 1 : 0;
    }
    return 0;
}

int fdtd_read_pgm_codes(const char *path, int nx, int ny, int upsample, int y0, int rows, uint8_t *dst)
{
    if (nx < 1 || ny < 1 || y0 < 0 || rows < 0 || y0 + rows > ny) {
        fdtd::set_error("fdtd_read_pgm_codes: rows outside the grid");
        return -3;
    }
    return fdtd::read_pgm_slice(path, nx, ny, upsample, y0, rows, dst);
}

int fdtd_material_coefficients(const fdtd_params *p, int ecoef, float sigma, float eps_s, float eps_inf,
                               float tau_d, float acd[3], double k[3])
{
    if (ecoef != FDTD_ECOEF_REFERENCE && ecoef != FDTD_ECOEF_CONSISTENT) {
        fdtd::set_error("unknown ecoef");
        return -3;
    }
    const fdtd::MatCoef m = fdtd::make_coef({sigma, eps_s, eps_inf, tau_d}, p->dt, p->eps0, ecoef);
    acd[0] = m.a;
    acd[1] = m.c;
    acd[2] = m.d;
    k[0] = m.k1;
    k[1] = m.k2;
    k[2] = m.k3;
    return 0;
}

float fdtd_rp_value(const fdtd_params *p, int x_face, float sigma, float eps_s, float eps_inf, float tau_d)
{
    return fdtd::rp_value(*p, {sigma, eps_s, eps_inf, tau_d}, x_face != 0);
}

int fdtd_params_load(const char *path, fdtd_params *out)
{
    fdtd_params p;
    int rc = fdtd::params_load(path, &p);
    if (rc != 0) return rc;
    rc = fdtd::setup_sources(&p);
    if (rc != 0) return rc;
    *out = p;
    return 0;
}

} // extern "C"
