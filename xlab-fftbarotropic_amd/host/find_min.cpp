// find_min.cpp -- last stage of the reference's pressure pipeline (find_min.cpp:67-101):
// for every file name on stdin, the 30 smallest values of that field with their grid positions, one
// "<ix> <iy> <value %.5e>" line each on stdout.  Host only: an O(N^2) scan of a field read through readField
// (lib/libfieldio.so).  The reference fixes the grid at compile time (configuration.hpp:18-21); here --npts / --xpts / --ypts.
//
// The ORDER of the 30 lines is part of the behaviour (downstream scripts take line 1 as "the" minimum position only after
// sorting; the raw order is what the reference prints): the reference keeps the first 30 values, then lets every smaller
// value evict the currently largest kept one (find_min.cpp:42-64), so the output is in eviction-slot order, unsorted.
// `Kept` below is that selection; tests/test_host_cpp.py pins the output line for line, ties included, to the reference's
// own find_min.cpp built into oracle/_ref.
#include <getopt.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "fieldio.hpp"

namespace {

struct Kept {
    std::vector<float> value;
    std::vector<size_t> where;
    size_t worst = 0;                                  // slot of the largest kept value: the FIRST such slot, as a strict '>' scan finds it

    explicit Kept(size_t n) : value(n), where(n) {}
    void rescan()
    {
        worst = 0;
        for (size_t s = 1; s < value.size(); ++s)
            if (value[s] > value[worst]) worst = s;
    }
    void select(const float *field, size_t count)
    {
        const size_t n = value.size();
        for (size_t s = 0; s < n; ++s) { value[s] = field[s]; where[s] = s; }
        rescan();
        for (size_t i = n; i < count; ++i) {
            if (!(field[i] < value[worst])) continue;  // strictly smaller values only: ties keep the earlier point
            value[worst] = field[i];
            where[worst] = i;
            rescan();
        }
    }
};

}  // namespace

int main(int argc, char *argv[])
{
    long xpts = 768, ypts = 768;                       // configuration.hpp:18
    const option longopts[] = {{"npts", required_argument, nullptr, 'n'}, {"xpts", required_argument, nullptr, 'x'},
                               {"ypts", required_argument, nullptr, 'y'}, {nullptr, 0, nullptr, 0}};
    for (int c; (c = getopt_long(argc, argv, "", longopts, nullptr)) != -1;) {
        switch (c) {
        case 'n': xpts = ypts = atol(optarg); break;
        case 'x': xpts = atol(optarg); break;
        case 'y': ypts = atol(optarg); break;
        default: fprintf(stderr, "usage: find_min.out [--npts N | --xpts NX --ypts NY] < list-of-files\n"); return 2;
        }
    }
    if (xpts < 1 || ypts < 1) { fprintf(stderr, "find_min: bad grid size\n"); return 2; }
    const size_t grids = (size_t)xpts * (size_t)ypts, wanted = 30;
    fprintf(stderr, "Entering find_min program.\n");
    if (wanted > grids) { fprintf(stderr, "Data size is %zu, but you request %zu numbers.\n", grids, wanted); return 1; }
    std::vector<float> field(grids);
    Kept kept(wanted);
    char line[1024];
    while (fgets(line, sizeof line, stdin)) {
        std::string name(line);
        while (!name.empty() && name.back() == '\n') name.pop_back();       // find_min.cpp:22-30
        readField(name.c_str(), field.data(), grids);
        fprintf(stderr, "File %s read.\n", name.c_str());
        kept.select(field.data(), grids);
        for (size_t s = 0; s < wanted; ++s)                                 // find_min.cpp:84-88
            printf("%zu %zu %.5e\n", kept.where[s] / (size_t)ypts, kept.where[s] % (size_t)ypts, kept.value[s]);
    }
    fprintf(stderr, "find_min program ends. Congrats!\n");
    return 0;
}
