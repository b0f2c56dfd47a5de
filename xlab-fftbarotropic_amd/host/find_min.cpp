// find_min.cpp -- post-processor of the reference's pressure pipeline (find_min.cpp:67-101):
// for every file name on stdin, the 30 smallest values of the field and their grid positions, one
// "<ix> <iy> <value %.5e>" line each on stdout, in the order the reference's selection leaves them.
// Host only (an O(N^2) scan of a field on disk); the field comes through readField (lib/libfieldio.so).
// The reference fixes the grid at compile time (configuration.hpp:18-21); here --npts / --xpts / --ypts.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <getopt.h>

#include "compat/fieldio.hpp"

static void trim(char *str)                      // find_min.cpp:22-30: strips trailing newlines
{
    size_t n = strlen(str);
    while (n > 0 && str[n - 1] == '\n') str[--n] = '\0';
}

static size_t find_max_pos(const float *data, size_t sz)      // find_min.cpp:32-40
{
    size_t max_i = 0;
    for (size_t i = 1; i < sz; ++i)
        if (data[i] > data[max_i]) max_i = i;
    return max_i;
}

// find_min.cpp:42-64: keep the first result_sz values, then replace the current maximum of the kept set by every
// smaller value met; the output order is the replacement order (not sorted), which the test pins
static void find_min_n(const float *data, size_t data_sz, float *result, size_t *result_pos, size_t result_sz)
{
    if (result_sz > data_sz) { fprintf(stderr, "Data size is %zu, but you request %zu numbers.\n", data_sz, result_sz); return; }
    for (size_t i = 0; i < result_sz; ++i) { result[i] = data[i]; result_pos[i] = i; }
    size_t max_i = find_max_pos(result, result_sz);
    for (size_t i = result_sz; i < data_sz; ++i) {
        if (data[i] < result[max_i]) {
            result[max_i] = data[i];
            result_pos[max_i] = i;
            max_i = find_max_pos(result, result_sz);
        }
    }
}

int main(int argc, char *argv[])
{
    int xpts = 768, ypts = 768;                   // configuration.hpp:18
    static const struct option lo[] = {{"npts", required_argument, 0, 1}, {"xpts", required_argument, 0, 2},
                                       {"ypts", required_argument, 0, 3}, {0, 0, 0, 0}};
    for (int c; (c = getopt_long(argc, argv, "", lo, nullptr)) != -1;) {
        if (c == 1) xpts = ypts = atoi(optarg);
        else if (c == 2) xpts = atoi(optarg);
        else if (c == 3) ypts = atoi(optarg);
        else { fprintf(stderr, "usage: find_min.out [--npts N | --xpts NX --ypts NY] < list-of-files\n"); return 2; }
    }
    if (xpts < 1 || ypts < 1) { fprintf(stderr, "find_min: bad grid size\n"); return 2; }
    const size_t grids = (size_t)xpts * ypts, min_n = 30;
    fprintf(stderr, "Entering find_min program.\n");
    float *data = (float *)malloc(sizeof(float) * grids), *mn = (float *)malloc(sizeof(float) * min_n);
    size_t *pos = (size_t *)malloc(sizeof(size_t) * min_n);
    if (!data || !mn || !pos) { fprintf(stderr, "find_min: out of memory\n"); return 1; }
    char filename[1024];
    while (fgets(filename, sizeof filename, stdin) != NULL) {
        trim(filename);
        readField(filename, data, grids);
        fprintf(stderr, "File %s read.\n", filename);
        find_min_n(data, grids, mn, pos, min_n);
        for (size_t i = 0; i < min_n && i < grids; ++i)
            fprintf(stdout, "%zu %zu %.5e\n", pos[i] / (size_t)ypts, pos[i] % (size_t)ypts, mn[i]);    // find_min.cpp:84-88
    }
    fprintf(stderr, "find_min program ends. Congrats!\n");
    free(data); free(mn); free(pos);
    return 0;
}
