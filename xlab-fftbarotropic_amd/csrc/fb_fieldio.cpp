// fb_fieldio.cpp -- field I/O of the drop-in boundary (host only, no HIP): fieldio.cpp:7-33.
//
// Two surfaces over one implementation:
//   * C ABI   fb_write_field / fb_read_field            (include/fftbaro.h; status added)
//   * C++     writeField / readField(const char*, float*, size_t)   -- the reference's own signatures
//             (fieldio.hpp:5-6), exported with the mangled names `_Z10writeFieldPKcPfm` / `_Z9readFieldPKcPfm`
//             that lib/libfieldio.so of the reference exports, so that objects compiled against the
//             reference's fieldio.hpp link against this library unchanged.
// Same bytes on disk (raw host-order float32, one fwrite/fread, no header) and the same stderr lines
// ("Output <file>", "<n> bytes read: <file>" -- the reference prints the ELEMENT count and calls it bytes,
// fieldio.cpp:32).  Differences, all on paths where the reference has undefined behaviour: a file that cannot be
// opened is reported (perror + status) instead of being handed to fwrite/fread as NULL (fieldio.cpp:9,23).
// Compiled twice: into libfftbaro.so and, alone, into lib/libfieldio.so (no GPU runtime needed to read a field).
#include <cstdio>
#include <string>

#include "../../include/fftbaro.h"

// fftbaro.hip provides it inside libfftbaro.so; the standalone libfieldio.so has no error string to set
extern "C" void fb_internal_set_error(const char *msg) __attribute__((weak));
static int io_fail(const std::string &msg) { if (fb_internal_set_error) fb_internal_set_error(msg.c_str()); return FB_EIO; }

extern "C" int fb_write_field(const char *filename, const float *data, size_t len)
{
    if (!filename || !data) { if (fb_internal_set_error) fb_internal_set_error("fb_write_field: NULL"); return FB_EINVAL; }
    FILE *f = fopen(filename, "wb");
    if (!f) { perror("Write field."); return io_fail(std::string("cannot open ") + filename); }
    const size_t n = fwrite(data, sizeof(float), len, f);
    fclose(f);
    fprintf(stderr, "Output %s\n", filename);                 // fieldio.cpp:18
    return n == len ? FB_OK : io_fail(std::string("short write: ") + filename);
}

extern "C" int fb_read_field(const char *filename, float *data, size_t len)
{
    if (!filename || !data) { if (fb_internal_set_error) fb_internal_set_error("fb_read_field: NULL"); return FB_EINVAL; }
    FILE *f = fopen(filename, "rb");
    if (!f) { perror("Read field."); return io_fail(std::string("cannot open ") + filename); }
    const size_t n = fread(data, sizeof(float), len, f);
    fclose(f);
    fprintf(stderr, "%d bytes read: %s\n", (int)n, filename);  // fieldio.cpp:32 (elements, labelled bytes)
    return n == len ? FB_OK : io_fail(std::string("short read: ") + filename);
}

// fieldio.hpp:5-6 -- void, like the reference's; failures are reported on stderr by the functions above
void writeField(const char *filename, float *data, size_t len) { (void)fb_write_field(filename, data, len); }
void readField(const char *filename, float *data, size_t len) { (void)fb_read_field(filename, data, len); }
