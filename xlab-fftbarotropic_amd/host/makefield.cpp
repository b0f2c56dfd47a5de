// makefield.cpp -- the initial-field generators as ONE host program on the C ABI (fb_make_field + fb_write_field).
//
// Mirrors the four reference programs makefield-elliptic-vortex.cpp:12-58, makefield-Kuo2004.cpp:30-41,
// makefield-gaussian.cpp:14-31 and makefield-const-vortex.cpp:14-38: each takes no arguments, builds GRIDS float32 and writes
// them to "<input>/<init_file>" = "input/initial_vorticity.bin" (configuration.hpp:39-41) through writeField.  Here the kind,
// the grid and the domain that configuration.hpp:15-18 fixes at compile time are run-time options:
//
//     makefield.out --kind elliptic|kuo2004|gaussian|const [--npts 768] [--lx 600000 --ly 600000] [-I input] [-i initial_vorticity.bin]
//
// Started under one of the reference's program names (a link named makefield-Kuo2004.out, makefield-elliptic-vortex.out,
// makefield-gaussian.out or makefield-const-vortex.out) it needs no --kind, so test/01-runtest/example.sh:3-10 and
// test/02-test_invert_pressure/example.sh:7 run unchanged on product binaries.  The values are bit-identical to the
// reference-built generators' (tests/test_host_cpp.py, hashes in tests/golden/ref_meta.json).
#include <getopt.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/fftbaro.h"

int main(int argc, char *args[])
{
    std::string input = "input", init_file = "initial_vorticity.bin", kind;       // configuration.hpp:39-41
    int npts = 768;                                                                // configuration.hpp:18
    float LX = 600000.0f, LY = 600000.0f;                                          // configuration.hpp:15-16
    const char *base = strrchr(args[0], '/');
    base = base ? base + 1 : args[0];
    if (!strncmp(base, "makefield-elliptic-vortex", 25)) kind = "elliptic";
    else if (!strncmp(base, "makefield-Kuo2004", 17)) kind = "kuo2004";
    else if (!strncmp(base, "makefield-gaussian", 18)) kind = "gaussian";
    else if (!strncmp(base, "makefield-const-vortex", 22)) kind = "const";
    static struct option lopts[] = {{"npts", 1, 0, 1}, {"lx", 1, 0, 2}, {"ly", 1, 0, 3}, {"kind", 1, 0, 4}, {0, 0, 0, 0}};
    int opt;
    while ((opt = getopt_long(argc, args, "I:i:", lopts, NULL)) != EOF) {
        switch (opt) {
        case 'I': input = optarg; break;                                           // the driver's letters (main.cpp:70-78)
        case 'i': init_file = optarg; break;
        case 1: npts = atoi(optarg); break;
        case 2: LX = (float)atof(optarg); break;
        case 3: LY = (float)atof(optarg); break;
        case 4: kind = optarg; break;
        default: return 2;
        }
    }
    if (kind.empty()) { fprintf(stderr, "usage: makefield.out --kind elliptic|kuo2004|gaussian|const [--npts N] [--lx LX --ly LY] [-I dir] [-i file]\n"); return 2; }
    if (npts < 1) { fprintf(stderr, "makefield: bad --npts\n"); return 2; }
    std::vector<float> vort((size_t)npts * npts);
    int rc = fb_make_field(kind.c_str(), npts, npts, LX, LY, vort.data());
    if (rc != FB_OK) { fprintf(stderr, "makefield: unknown --kind '%s' or bad grid (status %d)\n", kind.c_str(), rc); return 1; }
    const std::string file = input + "/" + init_file;
    rc = fb_write_field(file.c_str(), vort.data(), vort.size());                   // writeField: "Output <file>" on stderr (fieldio.cpp:18)
    return rc == FB_OK ? 0 : 1;
}
