// barotropic_main.cpp -- drop-in RK4 driver on the MI355X engine.
//
// Mirrors the surface of the reference drivers main.cpp:65-328 and main-shallow-water.cpp:72-349:
// same option letters (-I -O -i, plus -s -f of the source-forced variant), same output files
// (<O>/vort_src_input_step_N.bin, vort_step_N.bin, psi_step_N.bin, u_step_N.bin, v_step_N.bin),
// same ./log contents, same stdout lines, exit code 0.  The grid and model constants that
// configuration.hpp:10-36 fixes at compile time are run-time long options here.
// Stepping is the fused HIP path (fb_model_step / fb_slab_step); fields only leave HBM at record steps.
//
// Host I/O is off the critical path on every path:
//   record steps : record kernels on the compute stream -> D2H into pinned buffers on a copy stream -> a writer thread
//                  for the files and ./log, while the main thread goes on stepping (main.cpp:266-282 does all of it inline);
//   FIFO source  : a reader thread follows the producer's byte protocol (vorticity_source.cpp:112-133) ahead of the step
//                  loop into pinned buffers; a new source travels H2D on the copy stream and the compute stream waits for
//                  the copy's event only (main-shallow-water.cpp:304 reads and uploads inline).
//
// Multi-GPU (BASELINE configs 4 and 5; no reference counterpart): one process per GPU,
//     barotropic_main.out --world P --rank r --comm-file /shared/path --launch-token T ...   (same other options on every rank)
// rank 0 publishes the RCCL unique id through the comm file (host/comm_bootstrap.hpp: stale files of other launches are
// never taken), every rank reads its x rows of the initial field, steps through fb_slab_step (engine-driven RCCL all-to-all
// transposes) and writes its rows into the shared record files; rank 0 alone prints the step lines and writes ./log.  A
// FIFO source is read per rank from "<fifo>.<rank>" (vort_src_input.out --world P --rank r produces that rank's rows).
// --ranks-as-threads runs all P ranks as threads of ONE process on ONE GPU through the in-process transport: the
// rehearsal of the multi-rank host logic.
#include <fcntl.h>
#include <getopt.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fftbaro.h"
#include "comm_bootstrap.hpp"

static void must(int status, const char *what)
{
    if (status != FB_OK) { std::fprintf(stderr, "%s: %s (%s)\n", what, fb_strerror(status), fb_last_error()); std::exit(1); }
}

enum RECIPE_TYPE { SCRIPT, FIFO, EMPTY };            // vorticity_source.cpp:48-52

struct Config {
    std::string input = "input", output = "output", init_file = "initial_vorticity.bin", vort_src_filename, comm_file, token;
    int npts = 768, record_step = 100, total_steps = -1, start_step = 0;          // configuration.hpp:18,35,36
    float LX = 600000.0f, LY = 600000.0f, NU = 6.5f, dt = 3.0f;                    // configuration.hpp:15-17,34
    RECIPE_TYPE recipe_type = EMPTY;
    int world = 1, rank = 0; bool threads = false, fanout = false;
    long comm_max_age = 60, comm_timeout = 600;
    bool timing = true;                                                            // the [timing] summary on stderr (--no-timing: off)
    int record_buffers = 0;                                                        // sets of pinned record buffers: 0 = by size (two while a set is <= 1 GiB), 1, 2
    bool dump_grad = false, dump_dvortdt = false;                                  // the OUTPUT_GRAD_VORT / OUTPUT_DVORTDT blocks of main.cpp:156-162,170-176,229-235 as run-time options
};

// --fifo-fanout (multi-GPU, SURVEY.md section 8(e) "rank 0 reads, scatters x-slabs"): ONE producer that writes whole fields -- the
// reference's unmodified vort_src_input.out -- feeds every rank.  The lead rank reads `<fifo>` (flag byte per step, GRIDS float32 after
// a flag of 1: vorticity_source.cpp:112-133) and passes the same protocol on to `<fifo>.<r>`, each rank's x rows only; the ranks read
// their `<fifo>.<r>` as they do when per-rank producers feed them.  The launcher creates the FIFOs.
static void fifo_fanout(const Config cfg)
{
    const size_t slab = (size_t)(cfg.npts / cfg.world) * cfg.npts;
    FILE *in = fopen(cfg.vort_src_filename.c_str(), "rb");
    std::vector<FILE *> out(cfg.world, nullptr);
    bool opened = in != nullptr;
    if (!in) printf("ERROR: cannot open file [%s].\n", cfg.vort_src_filename.c_str());
    for (int r = 0; r < cfg.world && opened; ++r) {
        const std::string fn = cfg.vort_src_filename + "." + std::to_string(r);
        if ((out[r] = fopen(fn.c_str(), "wb")) == NULL) { printf("ERROR: cannot open file [%s].\n", fn.c_str()); opened = false; }
    }
    if (!opened) {
        // One FIFO of the fan-out is missing: the readers of those already opened get their EOF, and the launch ends here with a
        // failure instead of leaving ranks inside fopen/fread for ever (the reference has ONE reader, vorticity_source.cpp:121-124,
        // so it knows no such state).  _exit: this is a side thread of a process that is busy on the GPU.
        for (FILE *f : out) if (f) fclose(f);
        if (in) fclose(in);
        fflush(stdout); fflush(stderr);
        _exit(1);
    }
    std::vector<float> buf(slab * (size_t)cfg.world);
    char flag;
    while (fread(&flag, 1, 1, in) == 1) {
        if (((unsigned int)flag) != 1) { for (FILE *f : out) { fwrite(&flag, 1, 1, f); fflush(f); } continue; }
        // a record goes out only when ALL of it has arrived: a short one is passed on to no rank (every rank keeps the source it has,
        // finds EOF and says so), never to some of them -- the ranks must not integrate different forcings
        if (fread(buf.data(), sizeof(float), buf.size(), in) != buf.size()) { fprintf(stderr, "ERROR: Cannot read vorticity source input.\n"); fflush(stderr); break; }
        for (int r = 0; r < cfg.world; ++r) {                                          // the field is x-major: rank r's rows are the r-th piece of the record
            fwrite(&flag, 1, 1, out[r]);
            fwrite(buf.data() + (size_t)r * slab, sizeof(float), slab, out[r]);
            fflush(out[r]);
        }
    }
    for (FILE *f : out) fclose(f);                                                     // EOF for every rank ("No flag was detected")
    fclose(in);
}

// ---- the source: VortSrcRecipeReader<GRIDS> (vorticity_source.cpp:48-135) with the FIFO read ahead of the step loop ------------
// The byte protocol is the reference's: per step one flag byte, followed by `n` float32 when the flag is 1
// (vorticity_source.cpp:112-133).  A reader thread consumes it as fast as the producer writes and parks every new source in
// one of three pinned buffers; the step loop takes one entry per step, so what each step sees -- and what stderr says about
// it -- is what the inline fread of the reference would have seen.
struct SourceFeed {
    struct Entry { int status; int buf; };            // status: 0 flag 0, 1 new source in buf, -1 no flag (EOF), -2 short payload
    RECIPE_TYPE type = EMPTY; std::string filename; size_t n = 0; FILE *fifo = nullptr;
    float *pin[3] = {nullptr, nullptr, nullptr};
    int holds[3] = {0, 0, 0};                         // current source / queued / held by the record writer
    int cur = -1;                                     // buffer of the source in force (-1: zeros)
    std::deque<Entry> q; bool eof = false, quit = false;
    std::mutex mu; std::condition_variable cv; std::thread th;

    void init(RECIPE_TYPE t, const std::string &fn, size_t floats)
    {
        type = t; filename = fn; n = floats;
        if (type == SCRIPT) readScript();
        else if (type == FIFO) {
            if ((fifo = fopen(filename.c_str(), "rb")) == NULL) { printf("ERROR: cannot open file [%s].\n", filename.c_str()); return; }
            for (auto &p : pin) must(fb_malloc_host((void **)&p, n * sizeof(float)), "fb_malloc_host");
            th = std::thread([this] { run(); });
        }
    }
    int readScript()                                  // vorticity_source.cpp:100-110: only opens the file (unimplemented upstream)
    {
        FILE *fd = fopen(filename.c_str(), "r");
        if (fd == NULL) printf("ERROR: cannot open file [%s].\n", filename.c_str()); else fclose(fd);
        return 0;
    }
    void run()
    {
        for (;;) {
            char flag;
            if (fread(&flag, sizeof(char), 1, fifo) != 1) break;                     // EOF: every later step finds no flag either
            Entry e{0, -1};
            if (((unsigned int)flag) == 1) {
                std::unique_lock<std::mutex> lk(mu);
                int b = -1;
                cv.wait(lk, [&] { if (quit) return true; for (int i = 0; i < 3; ++i) if (holds[i] == 0) { b = i; return true; } return false; });
                if (quit) return;
                holds[b] = 1;
                lk.unlock();
                const bool ok = fread(pin[b], sizeof(float), n, fifo) == n;
                e = Entry{ok ? 1 : -2, b};
            }
            { std::lock_guard<std::mutex> lk(mu); q.push_back(e); }
            cv.notify_all();
            if (e.status == -2) break;
        }
        { std::lock_guard<std::mutex> lk(mu); eof = true; }
        cv.notify_all();
    }
    // one step's read (main-shallow-water.cpp:304); returns the entry, with the reference's stderr lines
    Entry read()
    {
        if (type == SCRIPT) { readScript(); return Entry{0, -1}; }
        if (type != FIFO) return Entry{0, -1};
        Entry e{-1, -1};
        if (fifo) {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return !q.empty() || eof; });
            if (!q.empty()) { e = q.front(); q.pop_front(); }
        }
        if (e.status == -1) { fprintf(stderr, "No flag was detected, assume flag = 0\n"); fflush(stderr); }
        else if (e.status == -2) { fprintf(stderr, "ERROR: Cannot read vorticity source input.\n"); fflush(stderr); release(e.buf); }
        else if (e.status == 1) fprintf(stderr, "New vorticity source was given.\n");
        else { fprintf(stderr, "No new vorticity source input was given.\n"); fflush(stderr); }
        return e;
    }
    void hold(int b) { if (b >= 0) { std::lock_guard<std::mutex> lk(mu); ++holds[b]; } }
    void release(int b) { if (b >= 0) { { std::lock_guard<std::mutex> lk(mu); --holds[b]; } cv.notify_all(); } }
    void make_current(int b) { const int old = cur; cur = b; release(old); }          // the queue's hold on b becomes the "current" hold
    // true: the reader is out of the FIFO and everything is released; false: it still sits in fread on an open FIFO (a producer
    // that outlives the run) -- the object, its stream and its buffers must then be left to process exit
    bool shutdown()
    {
        bool done;
        { std::lock_guard<std::mutex> lk(mu); quit = true; done = eof; }
        cv.notify_all();
        if (!th.joinable()) { if (fifo) fclose(fifo); return true; }
        if (!done) { th.detach(); return false; }
        th.join();
        fclose(fifo);
        for (auto p : pin) fb_free_host(p);
        return true;
    }
};

// ---- the record path: main.cpp:266-282 and the stage-0 dumps :181-222, written by a thread of its own ---------------------------
struct RecordWriter {
    // Up to TWO sets of pinned buffers, each with the event behind its D2H copies: the step loop hands a record over and goes on; it has
    // to wait only when BOTH sets are still with the writer, i.e. for the record before last.  With one set it waited for the previous
    // record's files at every record step, and one slow record -- measured: 85-96 ms for 336 MB where the average is 50 ms and the stretch
    // between two records 84 ms -- left the compute stream dry (1119-1155 against 1190 steps/s at 4096^2).
    struct Job { int step, set; const float *src; int src_buf; };
    std::thread th; std::mutex mu; std::condition_variable cv;
    std::deque<Job> jobs; bool writing = false, quit = false;
    int nsets = 1; bool set_free[2] = {true, true};
    void *e_copy[2] = {nullptr, nullptr}; float *h[2][7] = {{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}};
    // what a record step writes, in the reference's order (main.cpp:266-282, then the stage-0 dumps :156-235): name and buffer (-1 = vort_src)
    std::vector<std::pair<const char *, int> > items;
    SourceFeed *feed = nullptr;                                                        // vort_src as of the record step is held until written
    std::string output; FILE *log_fd = nullptr; size_t floats = 0;
    bool whole = true, lead = true; off_t off = 0;                                     // whole file (writeField) or this rank's byte range
    double busy_s = 0.0, slowest_s = 0.0; size_t bytes = 0;                            // time spent writing (total, slowest record) and what was written ([timing] summary)
    void start() { th = std::thread([this] { run(); }); }
    void run()
    {
        for (;;) {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [this] { return !jobs.empty() || quit; });
            if (jobs.empty() && quit) return;
            const Job job = jobs.front();
            jobs.pop_front();
            writing = true;
            lk.unlock();
            must(fb_event_synchronize(e_copy[job.set]), "record: wait for copies");
            const auto w0 = std::chrono::steady_clock::now();
            char fn[1024];
            for (size_t i = 0; i < items.size(); ++i) {                                // main.cpp:268-278, :156-235
                snprintf(fn, sizeof fn, "%s/%s_step_%d.bin", output.c_str(), items[i].first, job.step);
                const float *data = items[i].second < 0 ? job.src : h[job.set][items[i].second];
                if (whole) must(fb_write_field(fn, data, floats), "writeField");
                else {
                    const int fd = open(fn, O_WRONLY | O_CREAT, 0644);
                    const char *p = (const char *)data; size_t left = floats * sizeof(float); off_t o = off;
                    while (fd >= 0 && left) { const ssize_t k = pwrite(fd, p, left, o); if (k <= 0) break; p += k; left -= (size_t)k; o += k; }
                    if (fd < 0 || left) { perror("Write field."); std::exit(1); }
                    close(fd);
                    if (lead) fprintf(stderr, "Output %s\n", fn);
                }
                if (lead) { fprintf(log_fd, "%s\n", fn); fflush(log_fd); }
            }
            if (feed) feed->release(job.src_buf);
            const double this_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
            busy_s += this_s;
            if (this_s > slowest_s) slowest_s = this_s;
            bytes += items.size() * floats * sizeof(float);
            lk.lock();
            writing = false;
            set_free[job.set] = true;
            cv.notify_all();
        }
    }
    // a set of pinned buffers the writer is done with (blocks while every set is still queued or being written)
    int acquire()
    {
        std::unique_lock<std::mutex> lk(mu);
        int got = -1;
        cv.wait(lk, [&] { for (int i = 0; i < nsets; ++i) if (set_free[i]) { got = i; return true; } return false; });
        set_free[got] = false;
        return got;
    }
    void submit(const Job &j) { { std::lock_guard<std::mutex> lk(mu); jobs.push_back(j); } cv.notify_all(); }
    void wait_idle() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [this] { return jobs.empty() && !writing; }); }
    void stop() { wait_idle(); { std::lock_guard<std::mutex> lk(mu); quit = true; } cv.notify_all(); if (th.joinable()) th.join(); }
};

// ---- what the step loop needs from either model ------------------------------------------------------------------------------
struct Engine {
    virtual ~Engine() {}
    virtual void set_vort(const float *d) = 0;                                        // main.cpp:256
    virtual void get(float *d_vort, float *d_psi, float *d_u, float *d_v) = 0;        // record kernels, on the compute stream
    virtual void record(void *event) = 0;                                             // event behind what the compute stream holds
    virtual void wait(void *event) = 0;                                               // compute stream waits for it
    virtual void set_source(const float *d) = 0;                                      // main-shallow-water.cpp:304
    virtual void step() = 0;                                                          // main.cpp:286-317
    virtual void sync() = 0;
    // the optional stage-0 dumps of getDvortdt(debug) (main.cpp:156-162,170-176,229-235): dvortdx, dvortdy and
    // dvortdt = -u dvortdx - v dvortdy + vort_src of the CURRENT state (u, v as get() returned them; src NULL = zeros); any output may be NULL
    virtual void get_debug(float *d_dzdx, float *d_dzdy, float *d_dzdt, const float *d_u, const float *d_v, const float *d_src) = 0;
};
struct SingleEngine : Engine {
    fb_ctx *fop = nullptr; fb_model *model = nullptr; void *compute = nullptr;
    SingleEngine(const Config &cfg)
    {
        must(fb_create(&fop, cfg.npts, cfg.npts, cfg.LX, cfg.LY), "fb_create");
        must(fb_stream_create(&compute), "stream");
        must(fb_set_stream(fop, compute), "fb_set_stream");
        must(fb_model_create(&model, fop, cfg.NU, cfg.dt), "fb_model_create");
        npts = cfg.npts;
    }
    ~SingleEngine() { for (float *p : {d_spec, d_tmp, d_gx, d_gy}) if (p) fb_free(p); fb_model_destroy(model); fb_destroy(fop); fb_stream_destroy(compute); }
    void set_vort(const float *d) override { must(fb_model_set_vort(model, d), "fb_model_set_vort"); }
    void get(float *a, float *b, float *c, float *d) override
    {
        must(fb_model_get_vort(model, a), "fb_model_get_vort");
        must(fb_model_get_diag(model, b, c, d), "fb_model_get_diag");                   // functions of vort_c only: the reference's stage-0 values
    }
    void record(void *e) override { must(fb_event_record(e, compute), "record"); }
    void wait(void *e) override { must(fb_stream_wait_event(compute, e), "wait"); }
    void set_source(const float *d) override { must(fb_model_set_source(model, d), "fb_model_set_source"); }
    void step() override { must(fb_model_step(model, 1), "fb_model_step"); }
    void sync() override { must(fb_synchronize(fop), "sync"); }
    float *d_spec = nullptr, *d_tmp = nullptr, *d_gx = nullptr, *d_gy = nullptr; size_t nreal = 0;
    void get_debug(float *d_dzdx, float *d_dzdy, float *d_dzdt, const float *d_u, const float *d_v, const float *d_src) override
    {
        // the reference's own operator sequence on vort_c (main.cpp:151-168,225-227) through the operator entry points of the C ABI
        // (bit-exact standalone kernels): off the hot path, only on record steps and only when asked for
        if (!d_spec) {
            nreal = (size_t)npts * npts;
            const size_t spec = (size_t)npts * (npts / 2 + 1) * 2 * sizeof(float);
            must(fb_malloc((void **)&d_spec, spec), "fb_malloc"); must(fb_malloc((void **)&d_tmp, spec), "fb_malloc");
            must(fb_malloc((void **)&d_gx, nreal * sizeof(float)), "fb_malloc"); must(fb_malloc((void **)&d_gy, nreal * sizeof(float)), "fb_malloc");
        }
        must(fb_model_get_spectrum(model, d_spec), "fb_model_get_spectrum");
        float *gx = d_dzdx ? d_dzdx : d_gx, *gy = d_dzdy ? d_dzdy : d_gy;
        must(fb_gradx(fop, d_spec, d_tmp), "gradx"); must(fb_c2r(fop, d_tmp, gx, 1), "c2r");          // main.cpp:151,154
        must(fb_grady(fop, d_spec, d_tmp), "grady"); must(fb_c2r(fop, d_tmp, gy, 1), "c2r");          // main.cpp:165,168
        if (d_dzdt) must(fb_jacobian(fop, d_u, d_v, gx, gy, d_src, d_dzdt), "jacobian");               // main.cpp:225-227
    }
    int npts = 0;
};
struct SlabEngine : Engine {
    fb_slab *sl = nullptr;
    SlabEngine(const Config &cfg, int rank, void *hub)
    {
        if (!hub && rank == 0) fbcomm::prepare(cfg.comm_file);                         // before anything else: no rank may find a leftover
        must(fb_slab_create(&sl, cfg.npts, cfg.npts, cfg.LX, cfg.LY, cfg.NU, cfg.dt, rank, cfg.world), "fb_slab_create");
        if (hub) { must(fb_slab_connect_local(sl, hub), "fb_slab_connect_local"); return; }
        char id[FB_UNIQUE_ID_BYTES];
        if (rank == 0) {
            must(fb_slab_unique_id(id), "fb_slab_unique_id");
            if (!fbcomm::publish(cfg.comm_file, cfg.token, id, sizeof id)) { perror("comm file"); std::exit(1); }
        } else if (!fbcomm::await(cfg.comm_file, cfg.token, id, sizeof id, cfg.comm_timeout, cfg.comm_max_age)) {
            std::fprintf(stderr, "rank %d: no RCCL id of this launch in %s\n", rank, cfg.comm_file.c_str());
            std::exit(1);
        }
        must(fb_slab_connect_rccl(sl, id), "fb_slab_connect_rccl");
    }
    ~SlabEngine() { fb_slab_destroy(sl); }
    void set_vort(const float *d) override { must(fb_slab_set_vort_local(sl, d), "fb_slab_set_vort_local"); }
    void get(float *a, float *b, float *c, float *d) override
    {
        must(fb_slab_get_vort_local(sl, a), "fb_slab_get_vort_local");
        must(fb_slab_get_diag_local(sl, b, c, d), "fb_slab_get_diag_local");
    }
    void record(void *e) override { must(fb_slab_record_event(sl, e), "record"); }
    void wait(void *e) override { must(fb_slab_wait_event(sl, e), "wait"); }
    void set_source(const float *d) override { must(fb_slab_set_source_local(sl, d), "fb_slab_set_source_local"); }
    void step() override { must(fb_slab_step(sl, 1), "fb_slab_step"); }
    void sync() override { must(fb_slab_synchronize(sl), "sync"); }
    void get_debug(float *, float *, float *, const float *, const float *, const float *) override
    {
        std::fprintf(stderr, "--dump-grad-vort / --dump-dvortdt: one GPU only\n"); std::exit(2);      // (refused in main() already)
    }
};

// ---- one rank's run: the whole program when world == 1 --------------------------------------------------------------------------
static void run_rank(const Config &cfg, int rank, void *hub, FILE *log_fd, bool lead)
{
    const int N = cfg.npts, P = cfg.world;
    const size_t floats = (size_t)(N / P) * N;                                         // this rank's share of a field (all of it on one GPU)
    const off_t off = (off_t)rank * floats * sizeof(float);
    Engine *eng = P == 1 ? (Engine *)new SingleEngine(cfg) : (Engine *)new SlabEngine(cfg, rank, hub);
    void *copy = nullptr, *e_rec = nullptr, *e_h2d = nullptr, *e_src = nullptr;
    must(fb_stream_create(&copy), "stream");
    for (void **e : {&e_rec, &e_h2d, &e_src}) must(fb_event_create(e), "event");
    // record buffers 0..3 = vort, psi, u, v; 4, 5 = dvortdx, dvortdy (--dump-grad-vort); 6 = dvortdt (--dump-dvortdt)
    const bool use[7] = {true, true, true, true, cfg.dump_grad, cfg.dump_grad, cfg.dump_dvortdt};
    float *d_in = nullptr, *d_out[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    must(fb_malloc((void **)&d_in, floats * sizeof(float)), "fb_malloc");
    for (int i = 0; i < 7; ++i) if (use[i]) must(fb_malloc((void **)&d_out[i], floats * sizeof(float)), "fb_malloc");

    RecordWriter writer;
    size_t set_bytes = 0;
    for (int i = 0; i < 7; ++i) if (use[i]) set_bytes += floats * sizeof(float);
    writer.nsets = cfg.record_buffers == 1 || cfg.record_buffers == 2 ? cfg.record_buffers : (set_bytes <= ((size_t)1 << 30) ? 2 : 1);
    for (int b = 0; b < writer.nsets; ++b) {
        must(fb_event_create(&writer.e_copy[b]), "event");
        for (int i = 0; i < 7; ++i) if (use[i]) must(fb_malloc_host((void **)&writer.h[b][i], floats * sizeof(float)), "fb_malloc_host");
    }
    writer.items = {{"vort_src_input", -1}, {"vort", 0}};                              // main.cpp:268-278
    if (cfg.dump_grad) { writer.items.push_back({"dvortdx", 4}); writer.items.push_back({"dvortdy", 5}); }   // main.cpp:156-162,170-176
    writer.items.push_back({"psi", 1}); writer.items.push_back({"u", 2}); writer.items.push_back({"v", 3});   // main.cpp:181-222
    if (cfg.dump_dvortdt) writer.items.push_back({"dvortdt", 6});                      // main.cpp:229-235
    writer.output = cfg.output; writer.log_fd = log_fd; writer.floats = floats;
    writer.whole = P == 1; writer.lead = lead; writer.off = off;
    writer.start();
    int last_set = -1;                                                                 // the set the previous record's D2H copies went into
    std::vector<float> zeros(floats, 0.0f);                                            // vort_src before the first input (main.cpp:110 leaves it uninitialised)
    char filename[1024];

    snprintf(filename, sizeof filename, "%s/%s", cfg.input.c_str(), cfg.init_file.c_str());
    {   // readField (main.cpp:143-144) into a pinned buffer; a rank reads its rows of the file (fieldio.cpp:21-33 reads all of it)
        float *h0 = writer.h[0][0];
        if (P == 1) must(fb_read_field(filename, h0, floats), "readField");
        else {
            const int fd = open(filename, O_RDONLY);
            char *p = (char *)h0; size_t left = floats * sizeof(float); off_t o = off;
            while (fd >= 0 && left) { const ssize_t k = pread(fd, p, left, o); if (k <= 0) break; p += k; left -= (size_t)k; o += k; }
            if (fd < 0 || left) { perror("Read field."); std::exit(1); }
            close(fd);
            if (lead) fprintf(stderr, "%d bytes read: %s\n", (int)((size_t)N * N), filename);
        }
        must(fb_memcpy_h2d_async(copy, d_in, h0, floats * sizeof(float)), "h2d");
        must(fb_event_record(e_h2d, copy), "record");
        eng->wait(e_h2d);
    }
    SourceFeed *feedp = new SourceFeed;
    SourceFeed &feed = *feedp;
    const std::string fifo = (P == 1 || cfg.vort_src_filename.empty()) ? cfg.vort_src_filename : cfg.vort_src_filename + "." + std::to_string(rank);
    feed.init(cfg.recipe_type, fifo, floats);                                          // main-shallow-water.cpp:151-152
    writer.feed = &feed;
    if (lead) printf("Initialization complete.\n");
    eng->set_vort(d_in);                                                               // main.cpp:256
    eng->record(e_src);                                                                // d_in is free again behind this
    must(fb_event_synchronize(e_h2d), "sync");                                         // the pinned buffer goes back to the record path

    // [timing] ("add steps/s + GB/s summary"; the reference prints no timing, main.cpp:262-264).  The step loop is
    // never synchronised for it: time-stamped events on the compute stream bracket the stretches BETWEEN record steps -- a stretch ends
    // where the record branch begins, before the host waits for the writer, and the next one begins behind the record kernels -- so
    // their sum is the GPU time of the stepping alone; the host clock around the whole loop, writer included, gives the rate with records.
    const bool timing = cfg.timing && cfg.total_steps > cfg.start_step;
    std::vector<void *> t_beg, t_end;
    auto stamp = [&](std::vector<void *> &v) { if (!timing) return; void *e = nullptr; must(fb_event_create_timing(&e), "event"); eng->record(e); v.push_back(e); };
    eng->sync();
    const auto wall0 = std::chrono::steady_clock::now();
    int n_records = 0;
    double host_wait_s = 0.0, host_get_s = 0.0, host_copy_s = 0.0;                     // host time inside the record branch, by part
    auto since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); };
    stamp(t_beg);

    // The reference can be restarted from any vort_step_N.bin via -i, but always renumbers from 0
    // (SURVEY section 5); --start-step N continues the numbering and the source clock instead.
    for (int step = cfg.start_step; step < cfg.total_steps; ++step) {                  // main.cpp:260
        const bool record = (step % cfg.record_step) == 0;
        if (lead) { printf("# Step %d, time = %.2f", step, step * cfg.dt); if (record) printf(", record now!"); printf("\n"); }
        if (record) {                                                                  // main.cpp:266-282 and the stage-0 dumps :181-222
            stamp(t_end);
            ++n_records;
            auto h0 = std::chrono::steady_clock::now();
            const int set = writer.acquire();                                          // a set of pinned buffers the writer is done with
            host_wait_s += since(h0);
            if (last_set >= 0) eng->wait(writer.e_copy[last_set]);                     // device record buffers are free again: the previous record's copies have left them
            feed.hold(feed.cur);                                                       // vort_src as of this step (dumped BEFORE this step's read)
            RecordWriter::Job job{step, set, feed.cur >= 0 ? feed.pin[feed.cur] : zeros.data(), feed.cur};
            h0 = std::chrono::steady_clock::now();
            eng->get(d_out[0], d_out[1], d_out[2], d_out[3]);
            host_get_s += since(h0);
            if (cfg.dump_grad || cfg.dump_dvortdt) {
                // vort_src on the device: d_in holds the source in force once one has been uploaded (before that: zeros = NULL)
                eng->get_debug(d_out[4], d_out[5], d_out[6], d_out[2], d_out[3], feed.cur >= 0 ? d_in : nullptr);
                eng->record(e_src);                                                    // d_in may be overwritten behind this
            }
            eng->record(e_rec);
            must(fb_stream_wait_event(copy, e_rec), "wait");
            h0 = std::chrono::steady_clock::now();
            for (int i = 0; i < 7; ++i) if (use[i]) must(fb_memcpy_d2h_async(copy, writer.h[set][i], d_out[i], floats * sizeof(float)), "d2h");
            host_copy_s += since(h0);
            must(fb_event_record(writer.e_copy[set], copy), "record");
            last_set = set;
            writer.submit(job);
            stamp(t_beg);
        }
        if (cfg.recipe_type != EMPTY) {                                                // main-shallow-water.cpp:304
            const SourceFeed::Entry e = feed.read();
            if (e.status == 1) {
                must(fb_event_synchronize(e_h2d), "sync");                             // the buffer about to be given up has been uploaded
                feed.make_current(e.buf);
                must(fb_stream_wait_event(copy, e_src), "wait");                       // the previous set_source has read d_in
                must(fb_memcpy_h2d_async(copy, d_in, feed.pin[e.buf], floats * sizeof(float)), "h2d");
                must(fb_event_record(e_h2d, copy), "record");
                eng->wait(e_h2d);
                eng->set_source(d_in);
                eng->record(e_src);
            }
        }
        eng->step();                                                                   // main.cpp:286-317
    }
    stamp(t_end);
    eng->sync();                                                                       // the last step has run ...
    const auto compute_done = std::chrono::steady_clock::now();
    writer.stop();                                                                     // ... and now the last record's files are on their way
    const bool feed_done = feed.shutdown();
    must(fb_stream_synchronize(copy), "sync");
    if (timing) {
        const auto wall1 = std::chrono::steady_clock::now();
        const double wall_s = std::chrono::duration<double>(wall1 - wall0).count();
        const double tail_s = std::chrono::duration<double>(wall1 - compute_done).count();
        double gpu_ms = 0.0, rec_ms = 0.0;
        for (size_t k = 0; k < t_beg.size() && k < t_end.size(); ++k) { float ms = 0.0f; must(fb_event_elapsed_ms(t_beg[k], t_end[k], &ms), "elapsed"); gpu_ms += ms; }
        // what lies BETWEEN the stretches on the compute stream: a record branch from its first event to the one behind its record kernels,
        // i.e. those kernels plus whatever time the stream sat idle while the host was inside the branch (waiting for the writer, enqueueing)
        for (size_t k = 0; k + 1 < t_beg.size() && k < t_end.size(); ++k) { float ms = 0.0f; must(fb_event_elapsed_ms(t_end[k], t_beg[k + 1], &ms), "elapsed"); rec_ms += ms; }
        for (void *e : t_beg) fb_event_destroy(e);
        for (void *e : t_end) fb_event_destroy(e);
        if (lead) {
            const int steps = cfg.total_steps - cfg.start_step;
            const double rate = steps / (gpu_ms * 1e-3), gbs = 320.0 * N * (double)N * rate / 1e9, peak = 8000.0 * P;    // B_alg = 320 N^2 per step; HBM3E 8 TB/s per GPU
            fprintf(stderr, "[timing] %d RK4 steps, %d x %d grid, %d GPU%s: step loop without the record steps %.1f steps/s (%.4f ms/step) = %.0f GB/s by 320 N^2 B/step = %.3f of %.0f GB/s\n",
                    steps, N, N, P, P > 1 ? "s" : "", rate, gpu_ms / steps, gbs, gbs / peak, peak);
            fprintf(stderr, "[timing] with %d record steps (%.3f GB written%s): %.1f steps/s over %.3f s of wall time; the writer thread was busy %.3f s = %.2f GB/s to %s\n",
                    n_records, writer.bytes * (double)(P > 1 && !cfg.threads ? P : 1) / 1e9, P > 1 && !cfg.threads ? ", all ranks" : (P > 1 ? ", this rank" : ""), steps / wall_s, wall_s, writer.busy_s,
                    writer.busy_s > 0 ? writer.bytes / writer.busy_s / 1e9 : 0.0, cfg.output.c_str());
            // where the difference between the two rates goes, measured rather than guessed
            fprintf(stderr, "[timing] of those %.3f s: %.3f s stepping, %.3f s with a record step holding the compute stream (record kernels + idle while the host was in the record branch), "
                            "%.3f s between the last step's end and the last file (writer tail), %.3f s unaccounted (loop start-up, event bookkeeping)\n",
                    wall_s, gpu_ms * 1e-3, rec_ms * 1e-3, tail_s, wall_s - gpu_ms * 1e-3 - rec_ms * 1e-3 - tail_s);
            // The host runs a stretch ahead of the GPU and then waits here for the previous record's files; the compute stream only runs dry
            // when ONE record's files take longer than the stretch between two records minus its D2H copies (the slowest record tells).
            fprintf(stderr, "[timing] host time inside the %d record branches: %.3f s waiting for a free set of record buffers (%d set%s), %.3f s enqueueing the record kernels, %.3f s in the D2H copy calls; "
                            "slowest record %.3f s to write, a stretch between records is %.3f s of stepping\n",
                    n_records, host_wait_s, writer.nsets, writer.nsets > 1 ? "s" : "", host_get_s, host_copy_s, writer.slowest_s, n_records > 0 ? gpu_ms * 1e-3 * cfg.record_step / steps : 0.0);
            fflush(stderr);
        }
    }
    if (feed_done) delete feedp;
    fb_free(d_in); for (auto p : d_out) if (p) fb_free(p);
    for (int b = 0; b < 2; ++b) { for (int i = 0; i < 7; ++i) if (writer.h[b][i]) fb_free_host(writer.h[b][i]); if (writer.e_copy[b]) fb_event_destroy(writer.e_copy[b]); }
    delete eng;
    for (void *e : {e_rec, e_h2d, e_src}) fb_event_destroy(e);
    fb_stream_destroy(copy);
}

int main(int argc, char *args[])
{
    // configuration.hpp:10-41 defaults (NPTS = 768, configuration.hpp:18)
    Config cfg;
    static struct option lopts[] = {{"npts", 1, 0, 1}, {"lx", 1, 0, 2}, {"ly", 1, 0, 3}, {"nu", 1, 0, 4}, {"dt", 1, 0, 5},
                                    {"steps", 1, 0, 6}, {"record-step", 1, 0, 7}, {"start-step", 1, 0, 8},
                                    {"world", 1, 0, 9}, {"rank", 1, 0, 10}, {"comm-file", 1, 0, 11}, {"ranks-as-threads", 0, 0, 12},
                                    {"launch-token", 1, 0, 13}, {"comm-max-age", 1, 0, 14}, {"comm-timeout", 1, 0, 15}, {"fifo-fanout", 0, 0, 16},
                                    {"no-timing", 0, 0, 17}, {"dump-grad-vort", 0, 0, 18}, {"dump-dvortdt", 0, 0, 19}, {"record-buffers", 1, 0, 20},
                                    {0, 0, 0, 0}};
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
        case 13: cfg.token = optarg; break;              // the same string on every rank of one launch (job id, start time)
        case 14: cfg.comm_max_age = atol(optarg); break;
        case 15: cfg.comm_timeout = atol(optarg); break;
        case 16: cfg.fanout = true; break;
        case 17: cfg.timing = false; break;
        case 18: cfg.dump_grad = true; break;            // main.cpp:156-162,170-176 (#ifdef OUTPUT_GRAD_VORT; configuration.hpp:4-5 defines only OUTPUT_PSI and OUTPUT_WIND)
        case 19: cfg.dump_dvortdt = true; break;
        case 20: cfg.record_buffers = atoi(optarg); break;         // main.cpp:229-235 (#ifdef OUTPUT_DVORTDT)
        }
    }
    if (cfg.world < 1 || cfg.rank < 0 || cfg.rank >= cfg.world || (cfg.world > 1 && !cfg.threads && cfg.comm_file.empty()) ||
        cfg.token.size() >= fbcomm::TOKEN_BYTES) {
        fprintf(stderr, "usage: ... --world P --rank r --comm-file FILE [--launch-token T]   (or --world P --ranks-as-threads)\n"); return 2;
    }
    if ((cfg.dump_grad || cfg.dump_dvortdt) && cfg.world > 1) { fprintf(stderr, "--dump-grad-vort / --dump-dvortdt: one GPU only\n"); return 2; }
    if (cfg.total_steps < 0) cfg.total_steps = (int)(60 * 60 / cfg.dt);              // configuration.hpp:36
    float dx = 0, dy = 0;                                                            // printed before being set, main.cpp:89-90

    const bool lead = cfg.world == 1 || cfg.threads || cfg.rank == 0;                  // the rank that owns the banner, stdout's step lines and ./log
    if (cfg.world > 1 && !cfg.threads) {                                              // one process per GPU: rank r on device r mod (visible devices)
        int ndev = 0;
        must(fb_device_count(&ndev), "fb_device_count");
        if (ndev > 0) must(fb_set_device(cfg.rank % ndev), "fb_set_device");
    }
    if (lead) {
    printf("##### Model setting #####\n");
    printf("Initial file          : %s \n", cfg.init_file.c_str());
    printf("Input folder          : %s \n", cfg.input.c_str());
    printf("Output folder         : %s \n", cfg.output.c_str());
    printf("Length X              : %.3f [m]\n", cfg.LX);
    printf("Length Y              : %.3f [m]\n", cfg.LY);
    printf("Spatial Resolution dx : %.3f [m]\n", dx);
    printf("Spatial Resolution dy : %.3f [m]\n", dy);
    printf("Time Resolution dt    : %.3f [s]\n", cfg.dt);
    printf("#########################\n\n\n");
    printf("Start project.\n");
    }

    FILE *log_fd = lead ? fopen("log", "w") : fopen("/dev/null", "w");                // main.cpp:97
    if (log_fd == NULL) { perror("Open log file"); return 1; }
    std::thread fan;
    if (cfg.fanout && cfg.world > 1 && cfg.recipe_type == FIFO && lead) fan = std::thread(fifo_fanout, cfg);
    if (cfg.world > 1 && cfg.threads) {
        void *hub = nullptr;
        must(fb_local_hub_create(&hub, cfg.world), "fb_local_hub_create");
        std::vector<std::thread> ts;
        for (int r = 0; r < cfg.world; ++r) ts.emplace_back([&, r] { run_rank(cfg, r, hub, log_fd, r == 0); });
        for (auto &t : ts) t.join();
        fb_local_hub_destroy(hub);
    } else run_rank(cfg, cfg.rank, nullptr, log_fd, lead);
    if (fan.joinable()) fan.detach();                                                // (a producer that outlives the run keeps it inside fread)
    fclose(log_fd);
    if (lead) printf("Program ends. Congrats!\n");
    return 0;
}
