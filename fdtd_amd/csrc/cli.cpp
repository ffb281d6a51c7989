// `fdtd` -- command-line driver with the invocation shape of the reference
// (fdtd.cu:46-544): run from the directory that holds data/, no arguments
// needed; reads data/input_params, data/mat_specs_riken and the PGM slices of
// env_set_dir, steps the fields on the GPU through the C ABI and writes
// <output_dir>/{H,D,E}_field_{x,y,z}_%05d_z%05d.out like writeResults3d
// (fdtd.cu:1288-1494).  Errors print a message and exit(EXIT_FAILURE), the
// convention of the reference's CHECK macro (utils.h:5-13).
//
// Extensions (all optional):
//   --params PATH        parameter file (default data/input_params)
//   --specs PATH         material specs (default data/mat_specs_riken)
//   --mode 2d|3d         default: 2d when nz = 1
//   --ecoef reference|consistent
//   --first-slice N      1-based PGM index used for z = 0 (default 1)
//   --upsample U         repeat each PGM pixel U x U times
//   --no-materials       keep the fsigma/feps_s/feps_inf/ftau_d defaults everywhere
//   --output-every N     write fields every N steps (default 1, 0 = never)
//   --axis-lines         also write E_field_{x,y,z}_%05d.out: all nine fields along the three
//                        grid lines through every source (writeResults, fdtd.cu:1197-1285)
//   --no-planes          with --axis-lines: skip the per-plane files
//   --device N
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "fdtd_amd.h"

static void die(const char *what)
{
    std::printf("Error: %s: %s\n", what, fdtd_last_error());
    std::exit(EXIT_FAILURE);
}

// printParams, fdtd.cu:648-667
static void print_params(const fdtd_params &p)
{
    std::printf("Field size:                 %04dx%04dx%04d\n", p.nx, p.ny, p.nz);
    std::printf("Iterations count:           %d\n", p.iterations);
    std::printf("Input path:                 %s\n", p.input_path);
    std::printf("Output path:                %s\n", p.output_path);
    std::printf("Elements per wavelength:    %d\n", p.elements_per_wave);
    std::printf("Wave frequency:             %9.3E\n", p.wave_frequency);
    std::printf("Pulse width:                %9.3E\n", p.pulse_width);
    std::printf("Pulse modulation frequency: %9.3E\n", p.pulse_modulation_frequency);
    std::printf("Sources count:              %d\n", p.sources_count);
    for (int i = 0; i < p.sources_count; i++)
        std::printf("Source position:            %04dx%04dx%04d\n", p.sources[i * 3] + 1, p.sources[i * 3 + 1] + 1,
                    p.sources[i * 3 + 2] + 1);
    std::printf("Default sigma:              %9.3E\n", p.default_sigma);
    std::printf("Default eps_s:              %9.3E\n", p.default_eps_s);
    std::printf("Default eps_i:              %9.3E\n", p.default_eps_i);
    std::printf("Default tau_d:              %9.3E\n", p.default_tau_d);
    std::printf("\n");
}

// One field of one step: nz files of ny lines of nx values " % 9.3E"
// (fdtd.cu:1302-1318); `iteration` is the 0-based step index.
static void write_field(const fdtd_params &p, const char *family, char axis, int iteration,
                        const std::vector<float> &data)
{
    std::string line;
    char num[32];
    for (int iz = 0; iz < p.nz; iz++) {
        char path[2304];
        std::snprintf(path, sizeof path, "%s/%s_field_%c_%05d_z%05d.out", p.output_path, family, axis, iteration, iz);
        FILE *f = std::fopen(path, "w");
        if (!f) {
            std::printf("Couldn't open file %s\n", path);
            std::exit(EXIT_FAILURE);
        }
        for (int iy = 0; iy < p.ny; iy++) {
            line.clear();
            const float *row = &data[((size_t)iz * p.ny + iy) * p.nx];
            for (int ix = 0; ix < p.nx; ix++) {
                std::snprintf(num, sizeof num, " % 9.3E", row[ix]);
                line += num;
            }
            line += '\n';
            std::fwrite(line.data(), 1, line.size(), f);
        }
        std::fclose(f);
    }
}

// The per-plane files of one output iteration are formatted and written by a second thread (which
// hands each field to a thread of its own) while the main thread steps the fields on to the next output iteration (the reference hands each step's
// fields to copy threads and a writer thread, fdtd.cu:132-155, 253).  Two sets of host buffers:
// the main thread only waits when both are still being written.
class PlaneWriter {
public:
    struct Job {
        int iteration = 0;
        std::vector<float> field[9]; // empty: the 2-D mode does not hold that field (written as zeros)
        bool busy = false;
    };

    PlaneWriter(const fdtd_params &p) : p_(p), thread_([this] { run(); }) {}
    ~PlaneWriter()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        thread_.join();
    }
    // A set of buffers nobody is writing from.
    Job *acquire()
    {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return !jobs_[0].busy || !jobs_[1].busy; });
        Job *j = !jobs_[0].busy ? &jobs_[0] : &jobs_[1];
        j->busy = true;
        return j;
    }
    void submit(Job *j)
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            queue_.push_back(j);
        }
        cv_.notify_all();
    }
    void drain()
    {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return queue_.empty() && !jobs_[0].busy && !jobs_[1].busy; });
    }

private:
    void run()
    {
        static const char *family[3] = {"H", "D", "E"};
        const size_t cells = (size_t)p_.nx * p_.ny * p_.nz;
        std::vector<float> zeros;
        for (;;) {
            Job *j;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || !queue_.empty(); });
                if (queue_.empty()) return;
                j = queue_.front();
                queue_.erase(queue_.begin());
            }
            // the nine fields go to different files: one thread each (formatting " % 9.3E" is what
            // costs, about 50 ns per value; the reference has a writer thread per field family)
            for (int f = 0; f < 9; f++)
                if (j->field[f].empty() && zeros.empty()) zeros.assign(cells, 0.0f);
            std::thread workers[9];
            for (int f = 0; f < 9; f++)
                workers[f] = std::thread([&, f] {
                    write_field(p_, family[f / 3], "xyz"[f % 3], j->iteration, j->field[f].empty() ? zeros : j->field[f]);
                });
            for (auto &w : workers) w.join();
            {
                std::lock_guard<std::mutex> lk(m_);
                j->busy = false;
            }
            cv_.notify_all();
        }
    }

    fdtd_params p_;
    std::mutex m_;
    std::condition_variable cv_;
    Job jobs_[2];
    std::vector<Job *> queue_;
    bool stop_ = false;
    std::thread thread_;
};

// writeResults (fdtd.cu:1197-1285): for every source the x, y and z grid lines through it, one text
// line per cell: 1-based indices, then D, H, E (x y z each).  Read as windows, not whole fields.
static void write_axis_lines(fdtd_ctx *ctx, const fdtd_params &p, bool is2d, int iteration)
{
    static const int order[9] = {FDTD_DX, FDTD_DY, FDTD_DZ, FDTD_HX, FDTD_HY, FDTD_HZ, FDTD_EX, FDTD_EY, FDTD_EZ};
    const int extent[3] = {p.nx, p.ny, p.nz};
    for (int axis = 0; axis < 3; axis++) {
        char path[2304];
        std::snprintf(path, sizeof path, "%s/E_field_%c_%05d.out", p.output_path, "xyz"[axis], iteration + 1);
        FILE *f = std::fopen(path, "w");
        if (!f) {
            std::printf("Couldn't open file %s\n", path);
            std::exit(EXIT_FAILURE);
        }
        const int n = extent[axis];
        std::vector<float> line((size_t)9 * n);
        for (int isrc = 0; isrc < p.sources_count; isrc++) {
            const int sx = p.sources[isrc * 3], sy = p.sources[isrc * 3 + 1], sz = is2d ? 0 : p.sources[isrc * 3 + 2];
            for (int k = 0; k < 9; k++) {
                float *dst = &line[(size_t)k * n];
                const int field = order[k];
                const bool held = !is2d || field == FDTD_HX || field == FDTD_HY || field == FDTD_DZ || field == FDTD_EZ;
                if (!held) {
                    std::fill(dst, dst + n, 0.0f);
                    continue;
                }
                int rc = 0;
                if (axis == 0)
                    rc = fdtd_read_window(ctx, field, 0, (int64_t)sz * p.ny + sy, 1, 0, p.nx, dst);
                else if (axis == 1)
                    rc = fdtd_read_window(ctx, field, 0, (int64_t)sz * p.ny, p.ny, sx, 1, dst);
                else
                    for (int iz = 0; iz < p.nz && !rc; iz++)
                        rc = fdtd_read_window(ctx, field, 0, (int64_t)iz * p.ny + sy, 1, sx, 1, dst + iz);
                if (rc) die("reading a grid line");
            }
            for (int i = 0; i < n; i++) {
                const int ix = axis == 0 ? i : sx, iy = axis == 1 ? i : sy, iz = axis == 2 ? i : sz;
                std::fprintf(f, " %3d %3d %3d", ix + 1, iy + 1, iz + 1);
                for (int k = 0; k < 9; k++) std::fprintf(f, " % 9.3E", line[(size_t)k * n + i]);
                std::fprintf(f, "\n");
            }
        }
        std::fclose(f);
    }
}

int main(int argc, char **argv)
{
    std::string params_path = "data/input_params", specs_path = "data/mat_specs_riken";
    fdtd_options opt;
    fdtd_options_default(&opt);
    int first_slice = 1, upsample = 1, output_every = 1;
    bool materials = true, axis_lines = false, planes = true;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto value = [&]() -> const char * {
            if (i + 1 >= argc) {
                std::printf("Error: %s needs a value\n", a.c_str());
                std::exit(EXIT_FAILURE);
            }
            return argv[++i];
        };
        if (a == "--params") params_path = value();
        else if (a == "--specs") specs_path = value();
        else if (a == "--mode") opt.mode = std::strcmp(value(), "2d") == 0 ? FDTD_MODE_2D_TMZ : FDTD_MODE_3D;
        else if (a == "--ecoef")
            opt.ecoef = std::strcmp(value(), "consistent") == 0 ? FDTD_ECOEF_CONSISTENT : FDTD_ECOEF_REFERENCE;
        else if (a == "--first-slice") first_slice = std::atoi(value());
        else if (a == "--upsample") upsample = std::atoi(value());
        else if (a == "--no-materials") materials = false;
        else if (a == "--output-every") output_every = std::atoi(value());
        else if (a == "--device") opt.device = std::atoi(value());
        else if (a == "--axis-lines") axis_lines = true;
        else if (a == "--no-planes") planes = false;
        else {
            std::printf("Error: unknown argument %s\n", a.c_str());
            return EXIT_FAILURE;
        }
    }

    std::printf("Reading parameters...\n");
    fdtd_params p;
    if (fdtd_params_load(params_path.c_str(), &p)) die("reading parameters");
    print_params(p);

    std::printf("Initializing device field...\n");
    fdtd_ctx *ctx = nullptr;
    if (fdtd_create(&p, &opt, &ctx)) die("initializing device field");

    if (materials) {
        std::printf("Reading materials data...\n");
        if (fdtd_load_materials(ctx, specs_path.c_str(), p.input_path, first_slice, upsample))
            die("reading materials data");
    }
    std::printf("Initializing sources...\n\n");

    const bool is2d = opt.mode == FDTD_MODE_2D_TMZ || (opt.mode == FDTD_MODE_AUTO && p.nz == 1);
    const size_t cells = (size_t)p.nx * p.ny * p.nz;
    PlaneWriter writer(p);
    const auto t0 = std::chrono::steady_clock::now();
    for (int first = 0; first < p.iterations;) {
        // all the steps up to the next iteration that writes results go into one call, so that
        // the library can advance several time steps per pass over memory
        int i = p.iterations - 1;
        if (output_every > 0) i = std::min(i, (first + output_every - 1) / output_every * output_every);
        for (int k = first; k <= i; k++) std::printf("Running iteration %d\n", k + 1);
        if (fdtd_step(ctx, i - first + 1)) die("stepping");
        first = i + 1;
        if (output_every > 0 && i % output_every == 0) {
            std::printf("Writing results for iteration %d\n", i + 1);
            if (axis_lines) write_axis_lines(ctx, p, is2d, i);
            if (planes) {
                PlaneWriter::Job *job = writer.acquire();
                job->iteration = i;
                for (int field = 0; field < 9; field++) {
                    const bool held = !is2d || field == FDTD_HX || field == FDTD_HY || field == FDTD_DZ || field == FDTD_EZ;
                    if (!held) {
                        job->field[field].clear();
                        continue;
                    }
                    job->field[field].resize(cells);
                    if (fdtd_read_field(ctx, field, 0, job->field[field].data(), cells)) die("reading a field");
                }
                writer.submit(job); // formatted and written while the next iterations run
            }
        }
    }
    if (fdtd_sync(ctx)) die("synchronizing");
    writer.drain();
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("\n%d iterations, %.3f s, %.1f Mcell-updates/s (I/O included)\n", p.iterations, sec,
                (double)cells * p.iterations / sec / 1e6);
    fdtd_destroy(ctx);
    return 0;
}
