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

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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

int main(int argc, char *args[])
{
    // configuration.hpp:10-41 defaults (NPTS 768 is not a power of two: see DESIGN.md, "out of scope")
    std::string input = "input", output = "output", init_file = "initial_vorticity.bin", vort_src_filename;
    int npts = 1024, record_step = 100, total_steps = -1, start_step = 0;
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
    must(fb_model_create(&model, fop, NU, dt), "fb_model_create");
    float *d_field = nullptr, *d_psi = nullptr, *d_u = nullptr, *d_v = nullptr;
    must(fb_malloc((void **)&d_field, GRIDS * sizeof(float)), "fb_malloc");
    must(fb_malloc((void **)&d_psi, GRIDS * sizeof(float)), "fb_malloc");
    must(fb_malloc((void **)&d_u, GRIDS * sizeof(float)), "fb_malloc");
    must(fb_malloc((void **)&d_v, GRIDS * sizeof(float)), "fb_malloc");
    std::vector<float> host(GRIDS), vort_src(GRIDS, 0.0f);                           // vort_src defined as zeros (main.cpp:110 leaves it uninitialised)
    char filename[1024];

    snprintf(filename, sizeof filename, "%s/%s", input.c_str(), init_file.c_str());
    must(fb_read_field(filename, host.data(), GRIDS), "readField");                   // main.cpp:143-144
    must(fb_memcpy_h2d(fop, d_field, host.data(), GRIDS * sizeof(float)), "h2d");
    VortSrcReader vs_reader;
    vs_reader.init(recipe_type, vort_src_filename, &vort_src);                        // main-shallow-water.cpp:151-152
    printf("Initialization complete.\n");
    must(fb_model_set_vort(model, d_field), "fb_model_set_vort");                     // main.cpp:256

    auto dump = [&](const char *name, int step, float *dev) {
        snprintf(filename, sizeof filename, "%s/%s_step_%d.bin", output.c_str(), name, step);
        if (dev) must(fb_memcpy_d2h(fop, host.data(), dev, GRIDS * sizeof(float)), "d2h");
        must(fb_write_field(filename, dev ? host.data() : vort_src.data(), GRIDS), "writeField");
        fprintf(log_fd, "%s\n", filename); fflush(log_fd);
    };

    int record_flag = 0;
    // The reference can be restarted from any vort_step_N.bin via -i, but always renumbers from 0
    // (SURVEY section 5); --start-step N continues the numbering and the source clock instead.
    for (int step = start_step; step < total_steps; ++step) {                          // main.cpp:260
        printf("# Step %d, time = %.2f", step, step * dt);
        if ((record_flag = ((step % record_step) == 0))) printf(", record now!");
        printf("\n");
        if (record_flag) {                                                             // main.cpp:266-282 and the stage-0 dumps :181-222
            dump("vort_src_input", step, nullptr);
            must(fb_model_get_vort(model, d_field), "fb_model_get_vort");
            dump("vort", step, d_field);
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
        if (record_flag) {                        // psi/u/v are functions of vort_c only: same values as the reference's stage-0 dumps
            must(fb_model_get_diag(model, d_psi, d_u, d_v), "fb_model_get_diag");
            dump("psi", step, d_psi); dump("u", step, d_u); dump("v", step, d_v);
        }
        must(fb_model_step(model, 1), "fb_model_step");                                // main.cpp:286-317
    }
    must(fb_synchronize(fop), "sync");
    fclose(log_fd);
    fb_free(d_field); fb_free(d_psi); fb_free(d_u); fb_free(d_v);
    fb_model_destroy(model); fb_destroy(fop);
    printf("Program ends. Congrats!\n");
    return 0;
}
