// barotropic_main.cpp -- drop-in RK4 driver on the MI355X engine.
//
// Mirrors the surface of the reference drivers main.cpp:65-328 and main-shallow-water.cpp:72-349:
// same option letters (-I -O -i, plus -s -f of the source-forced variant), same output files
// (<O>/vort_src_input_step_N.bin, vort_step_N.bin, psi_step_N.bin, u_step_N.bin, v_step_N.bin),
// same ./log contents, same stdout lines, exit code 0.  The grid and model constants that
// configuration.hpp:10-36 fixes at compile time are run-time long options here.
// Stepping is the fused HIP path (fb_model_step); fields only leave HBM at record steps.
//
// Multi-GPU (BASELINE configs 4 and 5; no reference counterpart): one process per GPU,
//     barotropic_main.out --world P --rank r --comm-file /shared/path ...          (same other options on every rank)
// rank 0 writes the RCCL unique id to the comm file, the others wait for it; every rank reads its x rows of the initial field,
// steps through fb_slab_step (engine-driven RCCL all-to-all transposes) and writes its rows into the shared record files;
// rank 0 alone prints the step lines and writes ./log.  A FIFO source is read per rank from "<fifo>.<rank>"
// (vort_src_input.out --world P --rank r produces that rank's rows).  --ranks-as-threads runs all P ranks as threads
// of ONE process on ONE GPU through the in-process transport: the rehearsal of the multi-rank host logic.
#include <fcntl.h>
#include <getopt.h>
#include <sys/stat.h>
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

struct Config {
    std::string input = "input", output = "output", init_file = "initial_vorticity.bin", vort_src_filename, comm_file;
    int npts = 768, record_step = 100, total_steps = -1, start_step = 0;          // configuration.hpp:18,35,36
    float LX = 600000.0f, LY = 600000.0f, NU = 6.5f, dt = 3.0f;                    // configuration.hpp:15-17,34
    RECIPE_TYPE recipe_type = EMPTY;
    int world = 1, rank = 0; bool threads = false;
};

// ---- multi-GPU ranks -------------------------------------------------------------------------------------------
static bool pread_all(int fd, void *buf, size_t n, off_t off)
{
    char *p = (char *)buf;
    while (n) { const ssize_t k = pread(fd, p, n, off); if (k <= 0) return false; p += k; n -= (size_t)k; off += k; }
    return true;
}
static bool pwrite_all(int fd, const void *buf, size_t n, off_t off)
{
    const char *p = (const char *)buf;
    while (n) { const ssize_t k = pwrite(fd, p, n, off); if (k <= 0) return false; p += k; n -= (size_t)k; off += k; }
    return true;
}
// rank 0 publishes the RCCL id through a file (write to a temporary name, then rename: readers never see a partial id)
static void bootstrap_id(const Config &cfg, int rank, char *id)
{
    if (rank == 0) {
        must(fb_slab_unique_id(id), "fb_slab_unique_id");
        const std::string tmp = cfg.comm_file + ".tmp";
        FILE *f = fopen(tmp.c_str(), "wb");
        if (!f || fwrite(id, 1, FB_UNIQUE_ID_BYTES, f) != FB_UNIQUE_ID_BYTES) { perror("comm file"); std::exit(1); }
        fclose(f);
        if (rename(tmp.c_str(), cfg.comm_file.c_str()) != 0) { perror("comm file"); std::exit(1); }
        return;
    }
    for (int tries = 0; tries < 6000; ++tries) {                                    // up to 10 minutes
        FILE *f = fopen(cfg.comm_file.c_str(), "rb");
        if (f) { const size_t n = fread(id, 1, FB_UNIQUE_ID_BYTES, f); fclose(f); if (n == FB_UNIQUE_ID_BYTES) return; }
        usleep(100000);
    }
    std::fprintf(stderr, "rank %d: no RCCL id in %s\n", rank, cfg.comm_file.c_str());
    std::exit(1);
}

static void run_slab_rank(const Config &cfg, int rank, void *hub, FILE *log_fd)
{
    const int N = cfg.npts, P = cfg.world, XL = N / P;
    const size_t rows = (size_t)XL * N;                                             // this rank's share of a field
    const off_t off = (off_t)rank * rows * sizeof(float);
    fb_slab *sl = nullptr;
    must(fb_slab_create(&sl, N, N, cfg.LX, cfg.LY, cfg.NU, cfg.dt, rank, P), "fb_slab_create");
    if (hub) must(fb_slab_connect_local(sl, hub), "fb_slab_connect_local");
    else { char id[FB_UNIQUE_ID_BYTES]; bootstrap_id(cfg, rank, id); must(fb_slab_connect_rccl(sl, id), "fb_slab_connect_rccl"); }
    float *d_in = nullptr, *d_out[4] = {nullptr, nullptr, nullptr, nullptr};
    must(fb_malloc((void **)&d_in, rows * sizeof(float)), "fb_malloc");
    for (auto &p : d_out) must(fb_malloc((void **)&p, rows * sizeof(float)), "fb_malloc");
    std::vector<float> host(rows), vort_src(rows, 0.0f);
    char filename[1024];
    snprintf(filename, sizeof filename, "%s/%s", cfg.input.c_str(), cfg.init_file.c_str());
    {   // readField of this rank's rows (fieldio.cpp:21-33 reads the whole field; the bytes are the same)
        const int fd = open(filename, O_RDONLY);
        if (fd < 0 || !pread_all(fd, host.data(), rows * sizeof(float), off)) { perror("Read field."); std::exit(1); }
        close(fd);
        if (rank == 0) fprintf(stderr, "%d bytes read: %s\n", (int)((size_t)N * N), filename);
    }
    fb_ctx *hctx = nullptr;                                                         // a tiny context, only for the synchronous copies
    must(fb_create(&hctx, 64, 64, 1.0f, 1.0f), "fb_create");
    must(fb_memcpy_h2d(hctx, d_in, host.data(), rows * sizeof(float)), "h2d");
    VortSrcReader vs_reader;
    const std::string fifo = cfg.vort_src_filename.empty() ? std::string() : cfg.vort_src_filename + "." + std::to_string(rank);
    vs_reader.init(cfg.recipe_type, fifo, &vort_src);
    if (rank == 0) printf("Initialization complete.\n");
    must(fb_slab_set_vort_local(sl, d_in), "fb_slab_set_vort_local");
    must(fb_slab_synchronize(sl), "sync");
    const char *names[5] = {"vort_src_input", "vort", "psi", "u", "v"};
    for (int step = cfg.start_step; step < cfg.total_steps; ++step) {                // main.cpp:260
        const bool record = (step % cfg.record_step) == 0;
        if (rank == 0) { printf("# Step %d, time = %.2f", step, step * cfg.dt); if (record) printf(", record now!"); printf("\n"); }
        if (record) {                                                                // main.cpp:266-282, :181-222: every rank writes its rows
            must(fb_slab_get_vort_local(sl, d_out[0]), "fb_slab_get_vort_local");
            must(fb_slab_get_diag_local(sl, d_out[1], d_out[2], d_out[3]), "fb_slab_get_diag_local");
            must(fb_slab_synchronize(sl), "sync");
            for (int i = 0; i < 5; ++i) {
                snprintf(filename, sizeof filename, "%s/%s_step_%d.bin", cfg.output.c_str(), names[i], step);
                const float *src = vort_src.data();
                if (i > 0) { must(fb_memcpy_d2h(hctx, host.data(), d_out[i - 1], rows * sizeof(float)), "d2h"); src = host.data(); }
                const int fd = open(filename, O_WRONLY | O_CREAT, 0644);
                if (fd < 0 || !pwrite_all(fd, src, rows * sizeof(float), off)) { perror("Write field."); std::exit(1); }
                close(fd);
                if (rank == 0) { fprintf(stderr, "Output %s\n", filename); fprintf(log_fd, "%s\n", filename); fflush(log_fd); }
            }
        }
        if (cfg.recipe_type != EMPTY) {                                              // main-shallow-water.cpp:304, this rank's rows
            vs_reader.read(step * cfg.dt);
            if (vs_reader.fresh) {
                must(fb_memcpy_h2d(hctx, d_in, vort_src.data(), rows * sizeof(float)), "h2d");
                must(fb_slab_set_source_local(sl, d_in), "fb_slab_set_source_local");
                must(fb_slab_synchronize(sl), "sync");
                vs_reader.fresh = false;
            }
        }
        must(fb_slab_step(sl, 1), "fb_slab_step");                                    // main.cpp:286-317
    }
    must(fb_slab_synchronize(sl), "sync");
    fb_free(d_in); for (auto p : d_out) fb_free(p);
    fb_destroy(hctx);
    fb_slab_destroy(sl);
}

int main(int argc, char *args[])
{
    // configuration.hpp:10-41 defaults (NPTS = 768, configuration.hpp:18)
    Config cfg;
    static struct option lopts[] = {{"npts", 1, 0, 1}, {"lx", 1, 0, 2}, {"ly", 1, 0, 3}, {"nu", 1, 0, 4}, {"dt", 1, 0, 5},
                                    {"steps", 1, 0, 6}, {"record-step", 1, 0, 7}, {"start-step", 1, 0, 8},
                                    {"world", 1, 0, 9}, {"rank", 1, 0, 10}, {"comm-file", 1, 0, 11}, {"ranks-as-threads", 0, 0, 12}, {0, 0, 0, 0}};
    int opt;
    while ((opt = getopt_long(argc, args, "I:O:i:s:f:", lopts, NULL)) != EOF) {      // main.cpp:68-80, main-shallow-water.cpp:75-95
        switch (opt) {
        case 'I': cfg.input = optarg; break;
        case 'O': cfg.output = optarg; break;
        case 'i': cfg.init_file = optarg; break;
        case 's': cfg.vort_src_filename = optarg; cfg.recipe_type = SCRIPT; break;
        case 'f': cfg.vort_src_filename = optarg; cfg.recipe_type = FIFO; break;
        case 1: cfg.npts = atoi(optarg); break;
        case 2: cfg.LX = (float)atof(optarg); break;
        case 3: cfg.LY = (float)atof(optarg); break;
        case 4: cfg.NU = (float)atof(optarg); break;
        case 5: cfg.dt = (float)atof(optarg); break;
        case 6: cfg.total_steps = atoi(optarg); break;
        case 7: cfg.record_step = atoi(optarg); break;
        case 8: cfg.start_step = atoi(optarg); break;    // restart: -i vort_step_N.bin --start-step N keeps file numbering and source timing
        case 9: cfg.world = atoi(optarg); break;
        case 10: cfg.rank = atoi(optarg); break;
        case 11: cfg.comm_file = optarg; break;
        case 12: cfg.threads = true; break;
        }
    }
    std::string &input = cfg.input, &output = cfg.output, &init_file = cfg.init_file, &vort_src_filename = cfg.vort_src_filename;
    int &npts = cfg.npts, &record_step = cfg.record_step, &total_steps = cfg.total_steps, &start_step = cfg.start_step;
    float &LX = cfg.LX, &LY = cfg.LY, &NU = cfg.NU, &dt = cfg.dt;
    RECIPE_TYPE &recipe_type = cfg.recipe_type;
    if (cfg.world < 1 || cfg.rank < 0 || cfg.rank >= cfg.world || (cfg.world > 1 && !cfg.threads && cfg.comm_file.empty())) {
        fprintf(stderr, "usage: ... --world P --rank r --comm-file FILE   (or --world P --ranks-as-threads)\n"); return 2;
    }
    if (total_steps < 0) total_steps = (int)(60 * 60 / dt);                          // configuration.hpp:36
    const int XPTS = npts, YPTS = npts;
    const size_t GRIDS = (size_t)XPTS * YPTS;
    float dx = 0, dy = 0;                                                            // printed before being set, main.cpp:89-90

    const bool lead = cfg.world == 1 || cfg.threads || cfg.rank == 0;                  // the rank that owns the banner, stdout's step lines and ./log
    if (cfg.world > 1 && !cfg.threads) {                                              // one process per GPU: rank r on device r mod (visible devices)
        int ndev = 0;
        must(fb_device_count(&ndev), "fb_device_count");
        if (ndev > 0) must(fb_set_device(cfg.rank % ndev), "fb_set_device");
    }
    if (lead) {
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
    }

    FILE *log_fd = lead ? fopen("log", "w") : fopen("/dev/null", "w");                // main.cpp:97
    if (log_fd == NULL) { perror("Open log file"); return 1; }
    if (cfg.world > 1) {                                                              // multi-GPU: see run_slab_rank
        if (cfg.threads) {
            void *hub = nullptr;
            must(fb_local_hub_create(&hub, cfg.world), "fb_local_hub_create");
            std::vector<std::thread> ts;
            for (int r = 0; r < cfg.world; ++r) ts.emplace_back([&, r] { run_slab_rank(cfg, r, hub, log_fd); });
            for (auto &t : ts) t.join();
            fb_local_hub_destroy(hub);
        } else run_slab_rank(cfg, cfg.rank, nullptr, log_fd);
        fclose(log_fd);
        if (lead) printf("Program ends. Congrats!\n");
        return 0;
    }

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
