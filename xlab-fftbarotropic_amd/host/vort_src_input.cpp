// vort_src_input.cpp -- producer of the FIFO vorticity-source stream (mirror of the reference's
// vort_src_input.cpp:30-66): for step = 1 .. total_steps-1 one flag byte, and after a flag of 1 the
// GRIDS float32 of the new source: the Kuo2004 cake of 3e-3/duration switched on at beg_time
// (2 h) and zeros at end_time (5 h).  Grid, dt and step count are run-time options here.
//   vort_src_input.out --npts 1024 --dt 3 --steps 1200 [--beg-time 7200 --duration 10800] > fifo
#include <getopt.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/fftbaro.h"

int main(int argc, char *args[])
{
    int npts = 768, total_steps = -1, world = 1, rank = 0;          // --world P --rank r: only rank r's x rows are emitted (multi-GPU runs feed one FIFO per rank)
    float LX = 600000.0f, LY = 600000.0f, dt = 3.0f, duration = 3600.0 * 3.0, beg_time = 3600.0 * 2.0;   // :36-37
    static struct option lopts[] = {{"npts", 1, 0, 1}, {"lx", 1, 0, 2}, {"ly", 1, 0, 3}, {"dt", 1, 0, 5}, {"steps", 1, 0, 6},
                                    {"beg-time", 1, 0, 7}, {"duration", 1, 0, 8}, {"world", 1, 0, 9}, {"rank", 1, 0, 10}, {0, 0, 0, 0}};
    int opt;
    while ((opt = getopt_long(argc, args, "", lopts, NULL)) != EOF) {
        switch (opt) {
        case 1: npts = atoi(optarg); break;
        case 2: LX = (float)atof(optarg); break;
        case 3: LY = (float)atof(optarg); break;
        case 5: dt = (float)atof(optarg); break;
        case 6: total_steps = atoi(optarg); break;
        case 7: beg_time = (float)atof(optarg); break;
        case 8: duration = (float)atof(optarg); break;
        case 9: world = atoi(optarg); break;
        case 10: rank = atoi(optarg); break;
        }
    }
    if (total_steps < 0) total_steps = (int)(60 * 60 / dt);                   // configuration.hpp:36
    if (world < 1 || rank < 0 || rank >= world || npts % world) { fprintf(stderr, "vort_src_input: bad --world/--rank\n"); return 2; }
    const size_t GRIDS = (size_t)npts * npts, MINE = GRIDS / world, FIRST = MINE * rank;      // this rank's rows
    std::vector<float> vort(GRIDS, 0.0f);
    const float end_time = beg_time + duration;                              // :38
    const size_t beg_step = (size_t)(beg_time / dt), end_step = (size_t)(end_time / dt);   // :40-41
    char flag;
    for (size_t step = 1; step < (size_t)total_steps; ++step) {              // :43
        if (step == beg_step) {
            if (fb_make_source_kuo2004(npts, npts, LX, LY, duration, vort.data()) != FB_OK) return 1;   // :46
            flag = (char)1;
            fwrite(&flag, sizeof(char), 1, stdout); fwrite(vort.data() + FIRST, sizeof(float), MINE, stdout);
        } else if (step == end_step) {
            memset(vort.data(), 0, GRIDS * sizeof(float));                   // :53
            flag = (char)1;
            fwrite(&flag, sizeof(char), 1, stdout); fwrite(vort.data() + FIRST, sizeof(float), MINE, stdout);
        } else {
            flag = (char)0;
            fwrite(&flag, sizeof(char), 1, stdout);
        }
    }
    fprintf(stderr, "###### input program ends ######\n");
    return 0;
}
