// barotropic_main.cpp -- drop-in RK4 driver on the MI355X engine.
//
// Mirrors the surface of the reference drivers main.cpp:65-328 and main-shallow-water.cpp:72-349:
// same option letters (-I -O -i, plus -s -f of the source-forced variant), same output files
// (<O>/vort_src_input_step_N.bin, vort_step_N.bin, psi_step_N.bin, u_step_N.bin, v_step_N.bin),
// same ./log contents, same stdout lines, exit code 0.  The grid and model constants that
// configuration.hpp:10-36 fixes at compile time are run-time long options here.
// Stepping is the fused HIP path (fb_model_step); fields only leave HBM at record steps.
#include <getopt.h>
#include <unistd.h>

#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fftbaro.h"

static void must(int status, const char *what)
{
    if (status != FB_OK) { std::fprintf(stderr, "%s: %s (%s)\n", what, fb_strerror(status), fb_last_error()); std::exit(1); }
}

// VortSrcRecipeReader<GRIDS> restated (vorticity_source.cpp:48-135)
enum RECIPE_TYPE { SCRIPT, FIFO, EMPTY };
struct VortSrcReader {
    RECIPE_TYPE type = EMPTY; std::string filename; FILE *fifo = nullptr; std::vector<float> *vort_src = nullptr; bool fresh = false;
    void init(RECIPE_TYPE t, const std::string &fn, std::vector<float> *dst)
    {
        type = t; filename = fn; vort_src = dst;
        if (type == SCRIPT) readScript();
        else if (type == FIFO && (fifo = fopen(filename.c_str(), "rb")) == NULL) printf("ERROR: cannot open file [%s].\n", filename.c_str());
    }
    int read(float) { return type == FIFO ? readFIFO() : (type == SCRIPT ? readScript() : 0); }
    int readScript()                                   // vorticity_source.cpp:100-110: only opens the file (unimplemented upstream)
    {
        FILE *fd = fopen(filename.c_str(), "r");
        if (fd == NULL) printf("ERROR: cannot open file [%s].\n", filename.c_str()); else fclose(fd);
        return 0;
    }
    int readFIFO()                                     // vorticity_source.cpp:112-133
    {
        char new_flag;
        if (!fifo || fread(&new_flag, sizeof(char), 1, fifo) != 1) { fprintf(stderr, "No flag was detected, assume flag = 0\n"); fflush(stderr); return 1; }
        if (((unsigned int)new_flag) == 1) {
            if (fread(vort_src->data(), sizeof(float), vort_src->size(), fifo) != vort_src->size()) {
                fprintf(stderr, "ERROR: Cannot read vorticity source input.\n"); fflush(stderr); return 2;
            }
            fresh = true;
            fprintf(stderr, "New vorticity source was given.\n");
        } else { fprintf(stderr, "No new vorticity source input was given.\n"); fflush(stderr); }
        return 0;
    }
    ~VortSrcReader() { if (fifo) fclose(fifo); }
};

// Record path off the critical path: the main thread enqueues the record kernels and the D2H copies and
// goes on stepping; this thread waits for the copies and writes the files + ./log lines in the reference's order.
struct RecordWriter {
    std::thread th; std::mutex mu; std::condition_variable cv;
    bool has_job = false, quit = false; int step = 0;
    void *e_copy = nullptr; float *h[4] = {nullptr, nullptr, nullptr, nullptr}; std::vector<float> src_snapshot;
    std::string output; FILE *log_fd = nullptr; size_t grids = 0;
    void start() { th = std::thread([this] { run(); }); }
    void run()
    {
        for (;;) {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [this] { return has_job || quit; });
            if (!has_job && quit) return;
            const int st = step;
            lk.unlock();
            must(fb_event_synchronize(e_copy), "record: wait for copies");
            char fn[1024];
            const char *names[5] = {"vort_src_input", "vort", "psi", "u", "v"};
            for (int i = 0; i < 5; ++i) {                                              // main.cpp:268-278, :187-220
                snprintf(fn, sizeof fn, "%s/%s_step_%d.bin", output.c_str(), names[i], st);
                must(fb_write_field(fn, i == 0 ? src_snapshot.data() : h[i - 1], grids), "writeField");
                fprintf(log_fd, "%s\n", fn); fflush(log_fd);
            }
            lk.lock();
            has_job = false;
            cv.notify_all();
        }
    }
    void wait_idle() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [this] { return !has_job; }); }
    void submit(int st) { { std::lock_guard<std::mutex> lk(mu); step = st; has_job = true; } cv.notify_all(); }
    void stop() { wait_idle(); { std::lock_guard<std::mutex> lk(mu); quit = true; } cv.notify_all(); if (th.joinable()) th.join(); }
};

int main(int argc, char *args[])
{
    // configuration.hpp:10-41 defaults (NPTS = 768, configuration.hpp:18)
    std::string input = "input", output = "output", init_file = "initial_vorticity.bin", vort_src_filename;
    int npts = 768, record_step = 100, total_steps = -1, start_step = 0;
    float LX = 600000.0f, LY = 600000.0f, NU = 6.5f, dt = 3.0f;
    RECIPE_TYPE recipe_type = EMPTY;
    static struct option lopts[] = {{"npts", 1, 0, 1}, {"lx", 1, 0, 2}, {"ly", 1, 0, 3}, {"nu", 1, 0, 4}, {"dt", 1, 0, 5},
                                    {"steps", 1, 0, 6}, {"record-step", 1, 0, 7}, {"start-step", 1, 0, 8}, {0, 0, 0, 0}};
    int opt;
    while ((opt = getopt_long(argc, args, "I:O:i:s:f:", lopts, NULL)) != EOF) {      // main.cpp:68-80, main-shallow-water.cpp:75-95
        switch (opt) {
        case 'I': input = optarg; break;
        case 'O': output = optarg; break;
        case 'i': init_file = optarg; break;
        case 's': vort_src_filename = optarg; recipe_type = SCRIPT; break;
        case 'f': vort_src_filename = optarg; recipe_type = FIFO; break;
        case 1: npts = atoi(optarg); break;
        case 2: LX = (float)atof(optarg); break;
        case 3: LY = (float)atof(optarg); break;
        case 4: NU = (float)atof(optarg); break;
        case 5: dt = (float)atof(optarg); break;
        case 6: total_steps = atoi(optarg); break;
        case 7: record_step = atoi(optarg); break;
        case 8: start_step = atoi(optarg); break;    // restart: -i vort_step_N.bin --start-step N keeps file numbering and source timing
        }
    }
    if (total_steps < 0) total_steps = (int)(60 * 60 / dt);                          // configuration.hpp:36
    const int XPTS = npts, YPTS = npts;
    const size_t GRIDS = (size_t)XPTS * YPTS;
    float dx = 0, dy = 0;                                                            // printed before being set, main.cpp:89-90

    printf("##### Model setting #####\n");
    printf("Initial file          : %s \n", init_file.c_str());
    printf("Input folder          : %s \n", input.c_str());
    printf("Output folder         : %s \n", output.c_str());
    printf("Length X              : %.3f [m]\n", LX);
    printf("Length Y              : %.3f [m]\n", LY);
    printf("Spatial Resolution dx : %.3f [m]\n", dx);
    printf("Spatial Resolution dy : %.3f [m]\n", dy);
    printf("Time Resolution dt    : %.3f [s]\n", dt);
    printf("#########################\n\n\n");
    printf("Start project.\n");

    FILE *log_fd = fopen("log", "w");                                                 // main.cpp:97
    if (log_fd == NULL) { perror("Open log file"); return 1; }

    fb_ctx *fop = nullptr; fb_model *model = nullptr;
    must(fb_create(&fop, XPTS, YPTS, LX, LY), "fb_create");
    void *compute = nullptr, *copy = nullptr, *e_rec = nullptr, *e_copy = nullptr;
    must(fb_stream_create(&compute), "stream"); must(fb_stream_create(&copy), "stream");
    must(fb_event_create(&e_rec), "event"); must(fb_event_create(&e_copy), "event");
    must(fb_set_stream(fop, compute), "fb_set_stream");
    must(fb_model_create(&model, fop, NU, dt), "fb_model_create");
    float *d_field = nullptr, *d_vort = nullptr, *d_psi = nullptr, *d_u = nullptr, *d_v = nullptr;
    must(fb_malloc((void **)&d_field, GRIDS * sizeof(float)), "fb_malloc");
    must(fb_malloc((void **)&d_vort, GRIDS * sizeof(float)), "fb_malloc");
    must(fb_malloc((void **)&d_psi, GRIDS * sizeof(float)), "fb_malloc");
    must(fb_malloc((void **)&d_u, GRIDS * sizeof(float)), "fb_malloc");
    must(fb_malloc((void **)&d_v, GRIDS * sizeof(float)), "fb_malloc");
    RecordWriter writer;
    for (int i = 0; i < 4; ++i) must(fb_malloc_host((void **)&writer.h[i], GRIDS * sizeof(float)), "fb_malloc_host");
    writer.e_copy = e_copy; writer.output = output; writer.log_fd = log_fd; writer.grids = GRIDS; writer.src_snapshot.assign(GRIDS, 0.0f);
    writer.start();
    bool copies_pending = false;
    std::vector<float> host(GRIDS), vort_src(GRIDS, 0.0f);                           // vort_src defined as zeros (main.cpp:110 leaves it uninitialised)
    char filename[1024];

    snprintf(filename, sizeof filename, "%s/%s", input.c_str(), init_file.c_str());
    must(fb_read_field(filename, host.data(), GRIDS), "readField");                   // main.cpp:143-144
    must(fb_memcpy_h2d(fop, d_field, host.data(), GRIDS * sizeof(float)), "h2d");
    VortSrcReader vs_reader;
    vs_reader.init(recipe_type, vort_src_filename, &vort_src);                        // main-shallow-water.cpp:151-152
    printf("Initialization complete.\n");
    must(fb_model_set_vort(model, d_field), "fb_model_set_vort");                     // main.cpp:256

    int record_flag = 0;
    // The reference can be restarted from any vort_step_N.bin via -i, but always renumbers from 0
    // (SURVEY section 5); --start-step N continues the numbering and the source clock instead.
    for (int step = start_step; step < total_steps; ++step) {                          // main.cpp:260
        printf("# Step %d, time = %.2f", step, step * dt);
        if ((record_flag = ((step % record_step) == 0))) printf(", record now!");
        printf("\n");
        if (record_flag) {                                                             // main.cpp:266-282 and the stage-0 dumps :181-222
            writer.wait_idle();                                                        // pinned buffers and the snapshot are free again
            if (copies_pending) must(fb_stream_wait_event(compute, e_copy), "wait");  // device record buffers are free again
            writer.src_snapshot = vort_src;                                            // vort_src as of this step (dumped BEFORE this step's read)
            must(fb_model_get_vort(model, d_vort), "fb_model_get_vort");
            must(fb_model_get_diag(model, d_psi, d_u, d_v), "fb_model_get_diag");      // functions of vort_c only: the reference's stage-0 values
            must(fb_event_record(e_rec, compute), "record");
            must(fb_stream_wait_event(copy, e_rec), "wait");
            float *dev[4] = {d_vort, d_psi, d_u, d_v};
            for (int i = 0; i < 4; ++i) must(fb_memcpy_d2h_async(copy, writer.h[i], dev[i], GRIDS * sizeof(float)), "d2h");
            must(fb_event_record(e_copy, copy), "record");
            copies_pending = true;
            writer.submit(step);
        }
        if (recipe_type != EMPTY) {                                                    // main-shallow-water.cpp:304
            vs_reader.read(step * dt);
            if (vs_reader.fresh) {
                must(fb_memcpy_h2d(fop, d_field, vort_src.data(), GRIDS * sizeof(float)), "h2d");
                must(fb_model_set_source(model, d_field), "fb_model_set_source");
                must(fb_synchronize(fop), "sync");
                vs_reader.fresh = false;
            }
        }
        must(fb_model_step(model, 1), "fb_model_step");                                // main.cpp:286-317
    }
    writer.stop();
    must(fb_synchronize(fop), "sync");
    fclose(log_fd);
    fb_free(d_field); fb_free(d_vort); fb_free(d_psi); fb_free(d_u); fb_free(d_v);
    for (int i = 0; i < 4; ++i) fb_free_host(writer.h[i]);
    fb_model_destroy(model); fb_destroy(fop);
    fb_event_destroy(e_rec); fb_event_destroy(e_copy); fb_stream_destroy(copy); fb_stream_destroy(compute);
    printf("Program ends. Congrats!\n");
    return 0;
}
